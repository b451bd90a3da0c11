// lane_model.cpp - the HIP kernel's per-lane fp32 arithmetic (fpyv_amd/csrc/fpv_math.h) compiled
// for the host.
//
// TEST INFRASTRUCTURE ONLY.  It lets the CPU test-suite (a) bound the fp32-vs-float64 error of the
// kernel's formulation against the oracle before any GPU time is spent and (b) assert on the GPU
// box that the gfx950 kernel reproduces this arithmetic bit for bit (both sides are built with
// -ffp-contract=off and use explicit fmaf).  It is NOT a CPU fallback: nothing under fpyv_amd/
// loads this library, and the product fails with FPV_ENODEV when no GPU is present.
#include <stdint.h>
#include <string.h>

#include "../fpyv_amd/csrc/fpv_addr.h"
#include "../fpyv_amd/csrc/fpv_derive.h"
#include "../fpyv_amd/csrc/fpv_math.h"

extern "C" {

// state: [rows][ld] SoA like the product.  actions: [steps][n][4] (per_step) or [n][4] held.
// accel: [3][ld] or NULL; done/reward: [n] or NULL (last step's values).  Returns 0 or FPV_E*.
static FpvObjects g_objs = {0, 0, {0, 0, 0}, {0, 0, 0}, {}};
static float* g_pos_comp = nullptr;      // [6][ld] Kahan compensation rows (p, v) used by subsequent fpvl_run calls, or null
void fpvl_set_pos_comp(float* c) { g_pos_comp = c; }
// guidance override ([n][9] rotation matrices, [n] thrust forces) applied on every step of subsequent fpvl_run calls, or null
static const float* g_rot_over = nullptr;
static const float* g_thrust_over = nullptr;
void fpvl_set_override(const float* rot, const float* thrust) { g_rot_over = rot; g_thrust_over = thrust; }
// 1: subsequent fpvl_run calls evaluate the ground flag from all four motor heights even for the square X frame
// (the unit test of the two-height shortcut compares both)
static int g_general_motors = 0;
void fpvl_set_general_motors(int on) { g_general_motors = on; }

// object_list used by subsequent fpvl_run calls (count 0 = none)
void fpvl_set_objects(const fpv_objects_t* t)
{
    g_objs.count = t ? t->count : 0;
    for (int k = 0; k < g_objs.count; ++k) {
        g_objs.o[k].type = t->obj[k].type; g_objs.o[k].x = t->obj[k].x; g_objs.o[k].y = t->obj[k].y; g_objs.o[k].z = t->obj[k].z;
        g_objs.o[k].radius = t->obj[k].radius; g_objs.o[k].height = t->obj[k].height;
    }
}

int fpvl_run(const fpv_params_t* P, int64_t n, int steps, float* st, int64_t ld, const float* actions,
             int per_step, const float wind[3], float* accel, uint8_t* done, float* reward)
{
    FpvK K;
    const char* why = "";
    const int rc = fpv_derive_constants(P, &K, &why);
    if (rc != FPV_OK) return rc;
    // the two-height ground flag of the k-step kernels' X-frame loop (fpv_drone_step_lane<.., SQ = true>) is used
    // here whenever its precondition holds, unless a test asks for the four-height form
    const bool sq = K.motor_square && !(K.flags & FPV_FLAG_GROUND) && g_objs.count == 0 && !g_general_motors;
    fpv_objects_bounds(g_objs, K.contact_reach);          // the list-level test of the collision pass, as the kernel's host side fills it
    for (int64_t i = 0; i < n; ++i) {
        if (P->mode == FPV_MODE_DRONE) {
            FpvDroneState s;
            s.px = st[FPV_PX * ld + i]; s.py = st[FPV_PY * ld + i]; s.pz = st[FPV_PZ * ld + i];
            s.vx = st[FPV_VX * ld + i]; s.vy = st[FPV_VY * ld + i]; s.vz = st[FPV_VZ * ld + i];
            s.q.w = st[FPV_QW * ld + i]; s.q.x = st[FPV_QX * ld + i]; s.q.y = st[FPV_QY * ld + i]; s.q.z = st[FPV_QZ * ld + i];
            s.rx = st[FPV_RX * ld + i]; s.ry = st[FPV_RY * ld + i]; s.rz = st[FPV_RZ * ld + i];
            s.thrust = st[FPV_THRUST * ld + i];
            FpvStepOut o = {0, 0, 0, 0, false};
            for (int t = 0; t < steps; ++t) {
                const float* a = actions + ((per_step ? (int64_t)t * n : 0) + i) * 4;
                float kc[6] = {0, 0, 0, 0, 0, 0};
                if (g_pos_comp) { for (int k = 0; k < 6; ++k) kc[k] = g_pos_comp[k * ld + i]; }
                float* kp = g_pos_comp ? kc : nullptr;
                const float* ro = g_rot_over ? g_rot_over + i * 9 : nullptr;
                const float to = g_rot_over ? g_thrust_over[i] : 0.0f;
                if (g_objs.count > 0)
                    o = fpv_drone_step_lane<true>(K, s, a[0], a[1], a[2], a[3], wind[0], wind[1], wind[2], &g_objs, kp, ro, to);
                else if (sq)
                    o = fpv_drone_step_lane<false, true, true>(K, s, a[0], a[1], a[2], a[3], wind[0], wind[1], wind[2], nullptr, kp, ro, to);
                else
                    o = fpv_drone_step_lane<false>(K, s, a[0], a[1], a[2], a[3], wind[0], wind[1], wind[2], nullptr, kp, ro, to);
                if (g_pos_comp) {
                    const bool rst = (K.flags & FPV_FLAG_AUTO_RESET) && o.done;
                    for (int k = 0; k < 6; ++k) g_pos_comp[k * ld + i] = rst ? 0.0f : kc[k];
                }
                if ((K.flags & FPV_FLAG_AUTO_RESET) && o.done) fpv_drone_reset_lane(K, s);
            }
            st[FPV_PX * ld + i] = s.px; st[FPV_PY * ld + i] = s.py; st[FPV_PZ * ld + i] = s.pz;
            st[FPV_VX * ld + i] = s.vx; st[FPV_VY * ld + i] = s.vy; st[FPV_VZ * ld + i] = s.vz;
            st[FPV_QW * ld + i] = s.q.w; st[FPV_QX * ld + i] = s.q.x; st[FPV_QY * ld + i] = s.q.y; st[FPV_QZ * ld + i] = s.q.z;
            st[FPV_RX * ld + i] = s.rx; st[FPV_RY * ld + i] = s.ry; st[FPV_RZ * ld + i] = s.rz;
            st[FPV_THRUST * ld + i] = s.thrust;
            if (accel) { accel[i] = o.ax; accel[ld + i] = o.ay; accel[2 * ld + i] = o.az; }
            if (done) done[i] = o.done ? 1 : 0;
            if (reward) reward[i] = o.reward;
        } else {
            FpvRacerState s;
            s.px = st[FPV_PX * ld + i]; s.py = st[FPV_PY * ld + i]; s.pz = st[FPV_PZ * ld + i];
            s.vx = st[FPV_VX * ld + i]; s.vy = st[FPV_VY * ld + i]; s.vz = st[FPV_VZ * ld + i];
            s.q.w = st[FPV_QW * ld + i]; s.q.x = st[FPV_QX * ld + i]; s.q.y = st[FPV_QY * ld + i]; s.q.z = st[FPV_QZ * ld + i];
            for (int k = 0; k < 3; ++k) {
                s.w[k] = st[(FPV_R_OMEGA + k) * ld + i];
                s.ierr[k] = st[(FPV_R_IERR + k) * ld + i];
                s.lerr[k] = st[(FPV_R_LERR + k) * ld + i];
            }
            s.first = st[FPV_R_FIRST * ld + i];
            for (int k = 0; k < 3; ++k) { s.wlo[k] = st[(FPV_R_OMEGA_LO + k) * ld + i]; s.ilo[k] = st[(FPV_R_IERR_LO + k) * ld + i]; s.dflt[k] = st[(FPV_R_DFILT + k) * ld + i]; }
            float r = 0;
            for (int t = 0; t < steps; ++t) {
                const float* a = actions + ((per_step ? (int64_t)t * n : 0) + i) * 4;
                if (K.r_pid_variant)
                    r = K.r_wide ? fpv_racer_step_lane<true, 1>(K, s, a[0], a[1], a[2], a[3]) : fpv_racer_step_lane<false, 1>(K, s, a[0], a[1], a[2], a[3]);
                else
                    r = K.r_wide ? fpv_racer_step_lane<true, 0>(K, s, a[0], a[1], a[2], a[3]) : fpv_racer_step_lane<false, 0>(K, s, a[0], a[1], a[2], a[3]);
                if ((K.flags & FPV_FLAG_AUTO_RESET) && !(fabsf(s.pz) <= K.ceiling)) fpv_racer_reset_lane(s);
            }
            st[FPV_PX * ld + i] = s.px; st[FPV_PY * ld + i] = s.py; st[FPV_PZ * ld + i] = s.pz;
            st[FPV_VX * ld + i] = s.vx; st[FPV_VY * ld + i] = s.vy; st[FPV_VZ * ld + i] = s.vz;
            st[FPV_QW * ld + i] = s.q.w; st[FPV_QX * ld + i] = s.q.x; st[FPV_QY * ld + i] = s.q.y; st[FPV_QZ * ld + i] = s.q.z;
            for (int k = 0; k < 3; ++k) {
                st[(FPV_R_OMEGA + k) * ld + i] = s.w[k];
                st[(FPV_R_IERR + k) * ld + i] = s.ierr[k];
                st[(FPV_R_LERR + k) * ld + i] = s.lerr[k];
            }
            st[FPV_R_FIRST * ld + i] = s.first;
            if (K.r_wide) { for (int k = 0; k < 3; ++k) { st[(FPV_R_OMEGA_LO + k) * ld + i] = s.wlo[k]; st[(FPV_R_IERR_LO + k) * ld + i] = s.ilo[k]; } }
            if (K.r_pid_variant) { for (int k = 0; k < 3; ++k) st[(FPV_R_DFILT + k) * ld + i] = s.dflt[k]; }
            if (reward) reward[i] = r;
            if (done) done[i] = 0;
        }
    }
    return FPV_OK;
}

}  // extern "C"

// fp16-storage variant: pos [3][ld] fp32, sh = [5][ld] half2 pairs (uint32) followed by [ld] thrust halves;
// every step goes through the same unpack -> step -> pack (stochastic rounding keyed by
// fpv_round_seed(seed0, step0 + t): the buffer's rounding_seed and the handle's 64-bit step counter) as
// fpv_drone_step_h_kernel.
extern "C" int fpvl_run_h(const fpv_params_t* P, int64_t n, int steps, float* pos, uint32_t* sh, int64_t ld,
                          const float* actions, int per_step, const float wind[3], uint32_t seed0,
                          uint8_t* done, float* reward, uint64_t step0)
{
    FpvK K;
    const char* why = "";
    const int rc = fpv_derive_constants(P, &K, &why);
    if (rc != FPV_OK) return rc;
    uint16_t* thrust = reinterpret_cast<uint16_t*>(sh + FPV_HALF_PAIR_ROWS * ld);
    for (int64_t i = 0; i < n; ++i) {
        FpvStepOut o = {0, 0, 0, 0, false};
        for (int t = 0; t < steps; ++t) {
            FpvDroneState s;
            FpvHalfState h;
            s.px = pos[0 * ld + i]; s.py = pos[1 * ld + i]; s.pz = pos[2 * ld + i];
            for (int k = 0; k < FPV_HALF_PAIR_ROWS; ++k) h.w[k] = sh[k * ld + i];
            h.t = thrust[i];
            fpv_unpack_half(h, s);
            const float* a = actions + ((per_step ? (int64_t)t * n : 0) + i) * 4;
            o = fpv_drone_step_lane<false, true, false, false>(K, s, a[0], a[1], a[2], a[3], wind[0], wind[1], wind[2]);
            if ((K.flags & FPV_FLAG_AUTO_RESET) && o.done) fpv_drone_reset_lane(K, s);
            fpv_pack_half(s, fpv_round_seed(seed0, step0 + (uint64_t)t), K.noise.id_lo + (uint32_t)i, h);     // keyed by the GLOBAL drone id
            pos[0 * ld + i] = s.px; pos[1 * ld + i] = s.py; pos[2 * ld + i] = s.pz;
            for (int k = 0; k < FPV_HALF_PAIR_ROWS; ++k) sh[k * ld + i] = h.w[k];
            thrust[i] = h.t;
        }
        if (done) done[i] = o.done ? 1 : 0;
        if (reward) reward[i] = o.reward;
    }
    return FPV_OK;
}

// conversion helpers exposed for the unit tests
extern "C" uint16_t fpvl_f32_to_f16(float x, uint32_t rnd13, int stochastic) { return stochastic ? fpv_f32_to_f16_sr(x, rnd13) : fpv_f32_to_f16_rn(x); }
extern "C" float fpvl_f16_to_f32(uint16_t h) { return fpv_f16_to_f32(h); }
extern "C" uint16_t fpvl_f32_to_f16_rtz(float x) { return (uint16_t)fpv_pack_pair_rtz(x, 0.0f); }
// the storage words of ONE drone state (14 fp32 values in row order) as the fp16 kernels pack it: out[0..4] pair words, out[5] thrust half
extern "C" void fpvl_pack_state(const float st[14], uint32_t seed, uint32_t drone, uint32_t out[6])
{
    FpvDroneState s;
    s.px = st[0]; s.py = st[1]; s.pz = st[2]; s.vx = st[3]; s.vy = st[4]; s.vz = st[5];
    s.q.w = st[6]; s.q.x = st[7]; s.q.y = st[8]; s.q.z = st[9]; s.rx = st[10]; s.ry = st[11]; s.rz = st[12]; s.thrust = st[13];
    FpvHalfState h;
    fpv_pack_half(s, seed, drone, h);
    for (int k = 0; k < 5; ++k) out[k] = h.w[k];
    out[5] = h.t;
}

// a whole [14][ld] fp32 state <-> the storage of the fp16 kernels (pos [3][ld] fp32 + sh: five pair rows of uint32 and the
// row of thrust halves), drone i packed with rounding seed `seed` and global id id0 + i: what fpv_reset_kernel leaves behind
extern "C" void fpvl_pack_rows(const float* st, int64_t ld, int64_t n, uint32_t seed, uint32_t id0, float* pos, uint32_t* sh)
{
    uint16_t* thrust = reinterpret_cast<uint16_t*>(sh + FPV_HALF_PAIR_ROWS * ld);
    for (int64_t i = 0; i < n; ++i) {
        FpvDroneState s;
        s.px = st[0 * ld + i]; s.py = st[1 * ld + i]; s.pz = st[2 * ld + i]; s.vx = st[3 * ld + i]; s.vy = st[4 * ld + i]; s.vz = st[5 * ld + i];
        s.q.w = st[6 * ld + i]; s.q.x = st[7 * ld + i]; s.q.y = st[8 * ld + i]; s.q.z = st[9 * ld + i];
        s.rx = st[10 * ld + i]; s.ry = st[11 * ld + i]; s.rz = st[12 * ld + i]; s.thrust = st[13 * ld + i];
        FpvHalfState h;
        fpv_pack_half(s, seed, id0 + (uint32_t)i, h);
        pos[0 * ld + i] = s.px; pos[1 * ld + i] = s.py; pos[2 * ld + i] = s.pz;
        for (int k = 0; k < FPV_HALF_PAIR_ROWS; ++k) sh[k * ld + i] = h.w[k];
        thrust[i] = h.t;
    }
}
extern "C" void fpvl_unpack_rows(const float* pos, const uint32_t* sh, int64_t ld, int64_t n, float* st)
{
    const uint16_t* thrust = reinterpret_cast<const uint16_t*>(sh + FPV_HALF_PAIR_ROWS * ld);
    for (int64_t i = 0; i < n; ++i) {
        FpvHalfState h;
        for (int k = 0; k < FPV_HALF_PAIR_ROWS; ++k) h.w[k] = sh[k * ld + i];
        h.t = thrust[i];
        FpvDroneState s;
        fpv_unpack_half(h, s);
        st[0 * ld + i] = pos[0 * ld + i]; st[1 * ld + i] = pos[1 * ld + i]; st[2 * ld + i] = pos[2 * ld + i];
        st[3 * ld + i] = s.vx; st[4 * ld + i] = s.vy; st[5 * ld + i] = s.vz;
        st[6 * ld + i] = s.q.w; st[7 * ld + i] = s.q.x; st[8 * ld + i] = s.q.y; st[9 * ld + i] = s.q.z;
        st[10 * ld + i] = s.rx; st[11 * ld + i] = s.ry; st[12 * ld + i] = s.rz; st[13 * ld + i] = s.thrust;
    }
}

// stick-noise generator on the host: ns [4][ld] EMA state advanced `steps` times from step index
// step0; applied [steps][n][4] receives clip(base + gain * x_s) (base = 0 when base_actions is NULL)
extern "C" int fpvl_stick_noise(const fpv_params_t* P, int64_t n, int steps, float* ns, int64_t ld,
                                const float* base_actions, float* applied, uint64_t step0)
{
    FpvK K;
    const char* why = "";
    const int rc = fpv_derive_constants(P, &K, &why);
    if (rc != FPV_OK) return rc;
    for (int t = 0; t < steps; ++t)
        for (int64_t i = 0; i < n; ++i) {
            float s[4], a[4];
            for (int k = 0; k < 4; ++k) { s[k] = ns[k * ld + i]; a[k] = base_actions ? base_actions[((int64_t)t * n + i) * 4 + k] : 0.0f; }
            fpv_stick_noise(K.noise, step0 + (uint64_t)t, (uint64_t)i, fpv_normal_table_host, s, a);
            for (int k = 0; k < 4; ++k) { ns[k * ld + i] = s[k]; if (applied) applied[((int64_t)t * n + i) * 4 + k] = a[k]; }
        }
    return FPV_OK;
}

extern "C" void fpvl_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds)
{
    if (rounds == 7) fpv_philox4x32<7>(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
    else fpv_philox4x32<10>(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
extern "C" int fpvl_noise_philox_rounds(void) { return FPV_NOISE_PHILOX_ROUNDS; }
// the generator's inverse normal CDF on an array of 32-bit words
extern "C" void fpvl_normal_from_words(const uint32_t* w, float* z, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) z[i] = fpv_normal_from_word(w[i], fpv_normal_table_host);
}

// lane addressing (fpyv_amd/csrc/fpv_addr.h) exposed for the unit test of the 2^28-drone limit
extern "C" uint32_t fpvl_lane_offset(uint32_t i, uint32_t elem_bytes) { return fpv_lane_offset(i, elem_bytes); }
extern "C" int64_t fpvl_max_drones(void) { return FPV_MAX_DRONES; }
extern "C" void fpvl_sincos_wide(double x, double* s, double* c) { fpv_sincos_wide(x, s, c); }
extern "C" void fpvl_sincos_reduced(float x, float* s, float* c) { fpv_sincos_reduced(x, s, c); }
extern "C" uint32_t fpvl_round_seed(uint32_t base, uint64_t step) { return fpv_round_seed(base, step); }
// the first two members of Drone.step's return triple as the kernel forms them: q = (w, x, y, z), rates in deg/s
extern "C" void fpvl_return_matrices(const float q[4], const float rates[3], float rt[9], float gyro[9])
{
    FpvQuat Q; Q.w = q[0]; Q.x = q[1]; Q.y = q[2]; Q.z = q[3];
    fpv_return_matrices(Q, rates[0], rates[1], rates[2], rt, gyro);
}
// the reset kernel's attitude for a per-drone (roll, pitch, yaw) in degrees: q[4] = w, x, y, z
extern "C" void fpvl_quat_from_rpy_deg(float roll, float pitch, float yaw, float q[4])
{
    const FpvQuat r = fpv_quat_from_rpy_deg(roll, pitch, yaw);
    q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
}

// components.PID in the kernel's fp32 arithmetic over a sequence: k[8] = kP, kI, kD, dt, integral_clip,
// min_output, max_output, derivative_transition_rate; st[4] = integral, prev_derivative, previous_error, is_first
extern "C" void fpvl_pid_run(const double k[8], float st[4], int T, const float* current, const float* target, float* out)
{
    FpvPidK<float> P;
    memset(&P, 0, sizeof(P));
    P.dt = (float)k[3]; P.inv_dt = (float)(1.0 / k[3]);
    P.gain[0][0] = (float)k[0]; P.gain[0][1] = (float)k[1]; P.gain[0][2] = (float)k[2];
    P.integral_clip = (float)k[4]; P.min_output = (float)k[5]; P.max_output = (float)k[6];
    P.d_rate = (float)k[7]; P.om_d_rate = (float)(1.0 - k[7]);
    for (int t = 0; t < T; ++t) {
        out[t] = fpv_pid_axis<float, 1>(P, 0, current[t], target[t], st[3] != 0.0f, st[0], st[2], st[1]);
        st[3] = 0.0f;
    }
}

// rotation matrix -> unit quaternion (w,x,y,z) as the guidance override of the step kernel does it
extern "C" void fpvl_quat_from_rot(const float m[9], float q[4])
{
    const FpvQuat r = fpv_quat_from_rot(m);
    q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
}

