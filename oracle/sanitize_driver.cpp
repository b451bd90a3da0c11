// sanitize_driver.cpp - runs every entry point of the CPU builds (the float64 oracle, its SIMD-across-drones form and the
// host build of the kernel's lane arithmetic) under AddressSanitizer + UndefinedBehaviorSanitizer.
//
// TEST INFRASTRUCTURE ONLY (`make -C oracle asan`, run by tests/test_sanitizers.py).  Inputs are
// deterministic pseudo-random sticks on ragged batch sizes; the point is not the values but that no
// access leaves its buffer and no operation is undefined, in any variant.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../include/fpv_abi.h"
#include "fpv_oracle.h"

extern "C" {
int fpvs_drone_step_batch(const fpvo_params* P, int64_t n, int steps, double* state, const double* actions, int action_per_step,
                          const double wind[3], double* accel, uint8_t* done, int threads);
int fpvl_run(const fpv_params_t* P, int64_t n, int steps, float* st, int64_t ld, const float* actions, int per_step,
             const float wind[3], float* accel, uint8_t* done, float* reward);
int fpvl_run_h(const fpv_params_t* P, int64_t n, int steps, float* pos, uint32_t* sh, int64_t ld, const float* actions,
               int per_step, const float wind[3], uint32_t seed0, uint8_t* done, float* reward, uint64_t step0);
int fpvl_stick_noise(const fpv_params_t* P, int64_t n, int steps, float* ns, int64_t ld, const float* base_actions,
                     float* applied, uint64_t step0);
void fpvl_set_objects(const fpv_objects_t* t);
void fpvl_set_pos_comp(float* c);
void fpvl_set_override(const float* rot, const float* thrust);
void fpvl_pid_run(const double k[8], float st[4], int T, const float* current, const float* target, float* out);
void fpvl_sincos_wide(double x, double* s, double* c);
void fpvl_sincos_reduced(float x, float* s, float* c);
void fpvl_quat_from_rpy_deg(float roll, float pitch, float yaw, float q[4]);
}

static uint32_t g_lcg = 12345u;
static double urand() { g_lcg = g_lcg * 1664525u + 1013904223u; return (double)(g_lcg >> 8) / 16777216.0 * 2.0 - 1.0; }

static fpv_params_t abi_params(uint32_t mode, uint32_t flags)
{
    fpv_params_t P;
    memset(&P, 0, sizeof(P));
    P.struct_size = sizeof(P); P.mode = mode; P.flags = flags;
    P.dt = 1e-3; P.gravity = 9.81; P.mass = 0.75; P.max_rates = 200; P.rates_transition_rate = 0.7; P.thrust_transition_rate = 0.5;
    const double poly[4] = {-3.5693188139684359e-05, 9.0016725594999486e-03, 2.7025509193863934e-01, -4.6756286242420328e-02};
    memcpy(P.thrust_poly, poly, sizeof(poly));
    const double cd[3] = {1.8, 1.8, 1.2}, ar[3] = {0.015, 0.013, 0.078};
    memcpy(P.drag_coefficients, cd, sizeof(cd)); memcpy(P.cross_section_areas, ar, sizeof(ar));
    P.air_density = 1.2225;
    for (int m = 0; m < 4; ++m) { const double t = M_PI / 4 + m * M_PI / 2; P.motor_xy[m][0] = 0.127 * cos(t); P.motor_xy[m][1] = 0.127 * sin(t); }
    P.init_position[2] = 0.4; P.init_velocity[0] = 1.0; P.init_quat[0] = 1.0;
    P.ceiling = 1.5; P.goal[2] = 10.0;
    P.racer_mass = 0.5;
    for (int i = 0; i < 3; ++i) { P.racer_inertia[i] = 0.002016125; P.racer_pid[i][0] = 0.004; P.racer_pid[i][1] = 0.02; P.racer_pid[i][2] = 1e-6; }
    P.racer_velocity_damping = 0.9; P.motor_radius = 0.1; P.ground_spring = 100; P.ground_damping = 0;
    P.noise_transition = 0.1; P.noise_gain = 1.0; P.noise_seed = 7; P.drone_id_offset = 1000;
    P.pid_integral_clip = 0.05; P.pid_min_output = -0.004; P.pid_max_output = 0.006; P.pid_derivative_transition_rate = 0.3;
    return P;
}

static fpvo_params oracle_params(const fpv_params_t& A)
{
    fpvo_params P;
    memset(&P, 0, sizeof(P));
    P.dt = A.dt; P.gravity = A.gravity; P.mass = A.mass; P.max_rates = A.max_rates;
    P.rates_transition_rate = A.rates_transition_rate; P.thrust_transition_rate = A.thrust_transition_rate;
    memcpy(P.thrust_poly, A.thrust_poly, sizeof(P.thrust_poly));
    memcpy(P.drag_coefficients, A.drag_coefficients, sizeof(P.drag_coefficients));
    memcpy(P.cross_section_areas, A.cross_section_areas, sizeof(P.cross_section_areas));
    P.air_density = A.air_density;
    memcpy(P.motor_xy, A.motor_xy, sizeof(P.motor_xy));
    P.racer_mass = A.racer_mass; memcpy(P.racer_inertia, A.racer_inertia, sizeof(P.racer_inertia));
    memcpy(P.racer_pid, A.racer_pid, sizeof(P.racer_pid));
    P.racer_velocity_damping = A.racer_velocity_damping; P.motor_radius = A.motor_radius;
    P.ground_spring = A.ground_spring; P.ground_damping = A.ground_damping;
    P.pid_integral_clip = A.pid_integral_clip; P.pid_min_output = A.pid_min_output; P.pid_max_output = A.pid_max_output;
    P.pid_derivative_transition_rate = A.pid_derivative_transition_rate;
    return P;
}

int main()
{
    const float wind[3] = {0.3f, -0.2f, 0.1f};
    const double windd[3] = {0.3, -0.2, 0.1};
    fpv_objects_t objs;
    memset(&objs, 0, sizeof(objs));
    objs.count = 3;
    objs.obj[0] = {2, 0.3f, -0.2f, 0.9f, 0.35f, 0.0f};
    objs.obj[1] = {1, 1.2f, 0.4f, 0.0f, 0.5f, 1.1f};
    objs.obj[2] = {0, 0, 0, 0, 0, 0};
    // the reset kernel's attitude and the range-reduced sin / cos behind it, out to angles no reset will ever see
    for (int k = 0; k < 2000; ++k) {
        float q[4], sn, cs;
        const float big = (float)(urand() * (k % 7 == 0 ? 1.0e6 : 720.0));
        fpvl_quat_from_rpy_deg(big, (float)(urand() * 90.0), (float)(urand() * 720.0), q);
        fpvl_sincos_reduced(big, &sn, &cs);
        const double nn = (double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2] + (double)q[3] * q[3];
        if (!(fabs(nn - 1.0) < 1e-5) || !(fabs((double)sn * sn + (double)cs * cs - 1.0) < 1e-5)) { fprintf(stderr, "reset attitude: not unit\n"); return 3; }
    }
    const int sizes[4] = {1, 63, 257, 1000};
    for (int si = 0; si < 4; ++si) {
        const int64_t n = sizes[si], ld = (n + 63) / 64 * 64;
        const int steps = 37;
        std::vector<float> acts((size_t)steps * n * 4);
        std::vector<double> actsd(acts.size());
        for (size_t k = 0; k < acts.size(); ++k) { acts[k] = (float)urand(); if (k % 4 == 3) acts[k] = -0.3f - 0.7f * fabsf(acts[k]); actsd[k] = acts[k]; }
        // ---- lane model, drone mode: plain / auto-reset + ground / objects + Kahan / big-angle ----
        for (int variant = 0; variant < 4; ++variant) {
            fpv_params_t P = abi_params(FPV_MODE_DRONE, variant == 1 ? (uint32_t)(FPV_FLAG_AUTO_RESET | FPV_FLAG_GROUND) : variant == 2 ? (uint32_t)FPV_FLAG_AUTO_RESET : 0u);
            if (variant == 3) P.max_rates = 2.0e5;
            std::vector<float> st((size_t)FPV_DRONE_ROWS * ld, 0.0f), acc((size_t)3 * ld), rew((size_t)n), comp((size_t)6 * ld, 0.0f);
            std::vector<uint8_t> done((size_t)n);
            for (int64_t i = 0; i < n; ++i) { st[FPV_PZ * ld + i] = 0.4f + 0.3f * (float)urand(); st[FPV_VX * ld + i] = 1.0f; st[FPV_QW * ld + i] = 1.0f; }
            if (variant == 2) { fpvl_set_objects(&objs); fpvl_set_pos_comp(comp.data()); }
            if (fpvl_run(&P, n, steps, st.data(), ld, acts.data(), 1, wind, acc.data(), done.data(), rew.data()) != 0) return 2;
            if (fpvl_run(&P, n, 5, st.data(), ld, acts.data(), 0, wind, nullptr, nullptr, nullptr) != 0) return 2;      // held action, no outputs
            fpvl_set_objects(nullptr); fpvl_set_pos_comp(nullptr);
            for (int64_t i = 0; i < n; ++i) if (!(st[FPV_QW * ld + i] == st[FPV_QW * ld + i])) { fprintf(stderr, "NaN in lane model\n"); return 3; }
        }
        // ---- both builds, guidance override (rotation_matrix= / thrust_force=): every Shepperd branch, NaN = no override ----
        {
            fpv_params_t P = abi_params(FPV_MODE_DRONE, 0);
            const fpvo_params PO = oracle_params(P);
            std::vector<float> st((size_t)FPV_DRONE_ROWS * ld, 0.0f), rot((size_t)n * 9), thr((size_t)n);
            std::vector<uint8_t> done((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                st[FPV_PZ * ld + i] = 5.0f; st[FPV_QW * ld + i] = 1.0f;
                double E[9];
                fpvo_euler_zyx_matrix(3.2 * urand(), 1.5 * urand(), 3.2 * urand(), E);      // all attitudes: trace > 0 and the three diagonal branches
                for (int k = 0; k < 9; ++k) rot[(size_t)i * 9 + k] = (float)E[k];
                thr[(size_t)i] = (i % 5 == 4) ? NAN : (float)(7.0 + 3.0 * urand());
                double s19[FPVO_DRONE_STATE] = {0, 0, 5.0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0}, a3[3];
                uint8_t d = 0;
                fpvo_drone_step_guided(&PO, s19, &actsd[(size_t)i * 4], windd, (i % 5 == 4) ? nullptr : E, thr[(size_t)i], a3, &d);
            }
            fpvl_set_override(rot.data(), thr.data());
            if (fpvl_run(&P, n, 3, st.data(), ld, acts.data(), 1, wind, nullptr, done.data(), nullptr) != 0) return 2;
            fpvl_set_override(nullptr, nullptr);
            for (int64_t i = 0; i < n; ++i) if (!(st[FPV_QW * ld + i] == st[FPV_QW * ld + i])) { fprintf(stderr, "NaN in the override path\n"); return 3; }
        }
        // ---- lane model, fp16 storage ----
        {
            fpv_params_t P = abi_params(FPV_MODE_DRONE, FPV_FLAG_FP16_STATE | FPV_FLAG_AUTO_RESET);
            std::vector<float> pos((size_t)3 * ld, 0.0f), rew((size_t)n);
            std::vector<uint32_t> sh(((size_t)FPV_HALF_ROWS_TOTAL_HALVES * ld + 1) / 2, 0u);
            std::vector<uint8_t> done((size_t)n);
            for (int64_t i = 0; i < n; ++i) { pos[2 * ld + i] = 0.5f; sh[(size_t)1 * ld + i] = 0x3c000000u; }     // qw = 1.0 (high half of pair row 1)
            if (fpvl_run_h(&P, n, steps, pos.data(), sh.data(), ld, acts.data(), 1, wind, 9u, done.data(), rew.data(), 0xfffffffeull) != 0) return 2;       // crosses 2^32 steps
        }
        // ---- lane model, stick noise ----
        {
            fpv_params_t P = abi_params(FPV_MODE_DRONE, FPV_FLAG_STICK_NOISE);
            std::vector<float> ns((size_t)4 * ld, 0.0f), applied((size_t)steps * n * 4);
            if (fpvl_stick_noise(&P, n, steps, ns.data(), ld, acts.data(), applied.data(), 0xfffffffdull) != 0) return 2;     // crosses 2^32 steps
            if (fpvl_stick_noise(&P, n, steps, ns.data(), ld, nullptr, nullptr, 40ull) != 0) return 2;
        }
        // ---- lane model, racer: fp32 omega*dt / as written (float64) x both PID semantics ----
        for (int variant = 0; variant < 4; ++variant) {
            fpv_params_t P = abi_params(FPV_MODE_RACER, FPV_FLAG_AUTO_RESET);
            P.racer_omega_dt = variant & 1; P.racer_pid_variant = (variant >> 1) & 1; P.ceiling = 5e-4;
            std::vector<float> st((size_t)FPV_RACER_ROWS * ld, 0.0f), rew((size_t)n);
            std::vector<uint8_t> done((size_t)n);
            for (int64_t i = 0; i < n; ++i) { st[FPV_QW * ld + i] = 1.0f; st[FPV_R_FIRST * ld + i] = 1.0f; }
            std::vector<float> ra(acts);
            for (size_t k = 0; k < ra.size(); ++k) ra[k] = (k % 4 == 3) ? 4.0f + ra[k] : 40.0f * ra[k];
            if (fpvl_run(&P, n, steps, st.data(), ld, ra.data(), 1, wind, nullptr, done.data(), rew.data()) != 0) return 2;
        }
        // ---- oracle: drone (plain / ground / objects) and racer (both PID semantics), 1 and many threads ----
        for (int variant = 0; variant < 3; ++variant) {
            const fpv_params_t A = abi_params(FPV_MODE_DRONE, 0);
            fpvo_params P = oracle_params(A);
            P.ground = variant == 1;
            if (variant == 2) {
                P.n_objects = 3;
                for (int k = 0; k < 3; ++k) {
                    P.objects[k].type = objs.obj[k].type; P.objects[k].x = objs.obj[k].x; P.objects[k].y = objs.obj[k].y;
                    P.objects[k].z = objs.obj[k].z; P.objects[k].radius = objs.obj[k].radius; P.objects[k].height = objs.obj[k].height;
                }
            }
            std::vector<double> st((size_t)n * FPVO_DRONE_STATE, 0.0), acc((size_t)n * 3);
            std::vector<uint8_t> done((size_t)n);
            for (int64_t i = 0; i < n; ++i) { double* s = &st[(size_t)i * FPVO_DRONE_STATE]; s[2] = 0.4 + 0.3 * urand(); s[3] = 1.0; s[6] = s[10] = s[14] = 1.0; }
            fpvo_drone_step_batch(&P, n, steps, st.data(), actsd.data(), 1, windd, acc.data(), done.data(), 1);
            fpvo_drone_step_batch(&P, n, 3, st.data(), actsd.data(), 0, windd, nullptr, nullptr, 0);
            // the SIMD-across-drones form of the same step (fpv_oracle_simd.c): ragged tiles, per-step and held sticks; a
            // general object list must be refused
            std::vector<double> sv((size_t)n * FPVO_DRONE_STATE, 0.0);
            for (int64_t i = 0; i < n; ++i) { double* s = &sv[(size_t)i * FPVO_DRONE_STATE]; s[2] = 0.4 + 0.3 * urand(); s[3] = 1.0; s[6] = s[10] = s[14] = 1.0; }
            const int rc1 = fpvs_drone_step_batch(&P, n, steps, sv.data(), actsd.data(), 1, windd, acc.data(), done.data(), 1);
            const int rc2 = fpvs_drone_step_batch(&P, n, 3, sv.data(), actsd.data(), 0, windd, nullptr, nullptr, 0);
            if ((variant == 2) != (rc1 != 0) || rc1 != rc2) { fprintf(stderr, "fpvs_drone_step_batch: unexpected return codes %d %d\n", rc1, rc2); return 5; }
        }
        for (int variant = 0; variant < 4; ++variant) {
            const fpv_params_t A = abi_params(FPV_MODE_RACER, 0);
            fpvo_params P = oracle_params(A);
            P.racer_omega_dt = variant & 1; P.racer_pid_variant = (variant >> 1) & 1;
            std::vector<double> st((size_t)n * FPVO_RACER_STATE, 0.0);
            for (int64_t i = 0; i < n; ++i) { st[(size_t)i * FPVO_RACER_STATE + 9] = 1.0; st[(size_t)i * FPVO_RACER_STATE + 19] = 1.0; }
            fpvo_racer_step_batch(&P, n, steps, st.data(), actsd.data(), 1, 0);
            double q[4], R[9];
            fpvo_quat_wxyz_to_matrix(&st[6], R);
            fpvo_matrix_to_quat_wxyz(R, q);
        }
    }
    // ---- components.PID, both builds; the float64 sin/cos ----
    {
        const double k[8] = {0.8, 5.0, 0.3, 1e-3, 0.05, -1.0, 1.0, 0.5};
        double st[4] = {0, 0, 0, 1};
        float stf[4] = {0, 0, 0, 1};
        std::vector<float> cur(500), tgt(500), out(500);
        for (int t = 0; t < 500; ++t) { cur[t] = (float)(3.0 * urand()); tgt[t] = (float)urand(); (void)fpvo_pid_call(k, st, cur[t], tgt[t]); }
        fpvl_pid_run(k, stf, 500, cur.data(), tgt.data(), out.data());
        double s, c;
        for (int t = 0; t < 2000; ++t) { fpvl_sincos_wide(1.0e6 * urand(), &s, &c); if (!(fabs(s * s + c * c - 1.0) < 1e-12)) { fprintf(stderr, "sincos_wide off the unit circle\n"); return 4; } }
        double E[9];
        fpvo_euler_zyx_matrix(0.1, -0.2, 0.3, E);
    }
    printf("sanitizers: clean (%d threads max)\n", fpvo_max_threads());
    return 0;
}
