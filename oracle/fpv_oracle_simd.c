/*
 * fpv_oracle_simd.c - Drone.step in float64 with the DRONES as the vector axis (SIMD across drones).
 *
 * TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY - see fpv_oracle.h: nothing in the product path may include,
 * link or call this.  It exists so that bench.py's cpu_baseline is the host's best and not a scalar per-drone port
 * (VERDICT r4 #7): the same arithmetic as fpv_oracle.c's fpvo_drone_step (which follows
 * /root/reference/src/utils/components.py:220-248 line by line and is pinned by tests/golden), laid out so that the
 * compiler vectorises over drones: a tile of FPVS_TILE drones is transposed into structure-of-arrays locals, stepped
 * TIME-OUTER with every loop over the tile marked `#pragma omp simd` (no loop-carried dependence, no branch: the
 * ground contact of components.py:198-214 is arithmetic on masks), and transposed back.  sin / cos come from the vector
 * math library (libmvec, <= 4 ulp) through `omp declare simd`; everything else is IEEE + - * / sqrt in the oracle's
 * order, -ffp-contract=off.  tests/test_numpy_port.py holds it to 1e-12 of the scalar oracle.
 *
 * Covers what the benchmark workload uses: Drone.step with object_list = [] or [Ground] (fpvo_params.ground), wind,
 * no guidance override, no general object list (n_objects must be 0 - the call refuses otherwise).
 */
#include "fpv_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* glibc attaches its vector variants (libmvec: _ZGV{b,c,d,e}N{2,4,8}v_sin / _cos) to sin / cos only under -ffast-math.  The
 * same `omp declare simd` on two names of our own that are BOUND to the libm symbols lets the loop vectoriser call them
 * without giving up IEEE arithmetic everywhere else - and keeps gcc from fusing sin(x), cos(x) into a scalar sincos call,
 * which has no vector variant. */
#if defined(__x86_64__) && defined(__GLIBC__) && !defined(FPVS_NO_LIBMVEC)
#pragma omp declare simd notinbranch
extern double fpvs_sin(double) __asm__("sin") __attribute__((const, nothrow, leaf));
#pragma omp declare simd notinbranch
extern double fpvs_cos(double) __asm__("cos") __attribute__((const, nothrow, leaf));
#else
#define fpvs_sin sin
#define fpvs_cos cos
#endif

#define FPVS_TILE 256
#define DEG2RAD (M_PI / 180.0)

int fpvs_tile(void) { return FPVS_TILE; }

static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* one tile: state [m][19] (m <= FPVS_TILE drones, AoS as fpv_oracle.h defines it), actions [steps][n_total][4] starting at
 * this tile's first drone (row stride n_total * 4 doubles; 0 = the same [m][4] batch every step) */
static void tile_run(const fpvo_params* P, int m, int steps, double* state, const double* actions, int64_t action_step_stride,
                     const double wind[3], double* accel, uint8_t* done)
{
    double S[FPVO_DRONE_STATE][FPVS_TILE] __attribute__((aligned(64)));
    double A[4][FPVS_TILE] __attribute__((aligned(64)));
    double AX[FPVS_TILE] __attribute__((aligned(64))), AY[FPVS_TILE] __attribute__((aligned(64))), AZ[FPVS_TILE] __attribute__((aligned(64)));
    double DN[FPVS_TILE] __attribute__((aligned(64)));
    for (int k = 0; k < FPVO_DRONE_STATE; ++k)
        for (int l = 0; l < FPVS_TILE; ++l)
            S[k][l] = l < m ? state[(int64_t)l * FPVO_DRONE_STATE + k] : (k == 6 || k == 10 || k == 14 ? 1.0 : 0.0);   /* padding lanes: identity attitude */
    double* px = S[0]; double* py = S[1]; double* pz = S[2];
    double* vx = S[3]; double* vy = S[4]; double* vz = S[5];
    double* r0 = S[6]; double* r1 = S[7]; double* r2 = S[8]; double* r3 = S[9]; double* r4 = S[10]; double* r5 = S[11];
    double* r6 = S[12]; double* r7 = S[13]; double* r8 = S[14];
    double* qx = S[15]; double* qy = S[16]; double* qz = S[17]; double* pt = S[18];

    const double maxr = P->max_rates, rtr = P->rates_transition_rate, ttr = P->thrust_transition_rate, dt = P->dt;
    const double c0 = P->thrust_poly[0], c1 = P->thrust_poly[1], c2 = P->thrust_poly[2], c3 = P->thrust_poly[3];
    const double k0 = -0.5 * P->drag_coefficients[0] * P->air_density * P->cross_section_areas[0];
    const double k1 = -0.5 * P->drag_coefficients[1] * P->air_density * P->cross_section_areas[1];
    const double k2 = -0.5 * P->drag_coefficients[2] * P->air_density * P->cross_section_areas[2];
    const double gz = -P->gravity * P->mass, mass = P->mass;
    const double wx = wind[0], wy = wind[1], wz = wind[2];
    const double mrad = P->motor_radius, ksp = P->ground_spring, kdm = P->ground_damping;
    const double groundf = P->ground ? 1.0 : 0.0;
    const double m0x = P->motor_xy[0][0], m0y = P->motor_xy[0][1], m1x = P->motor_xy[1][0], m1y = P->motor_xy[1][1];
    const double m2x = P->motor_xy[2][0], m2y = P->motor_xy[2][1], m3x = P->motor_xy[3][0], m3y = P->motor_xy[3][1];

    for (int t = 0; t < steps; ++t) {
        const double* at = actions + (int64_t)t * action_step_stride;
        for (int l = 0; l < m; ++l) { A[0][l] = at[4 * l + 0]; A[1][l] = at[4 * l + 1]; A[2][l] = at[4 * l + 2]; A[3][l] = at[4 * l + 3]; }
        for (int l = m; l < FPVS_TILE; ++l) A[0][l] = A[1][l] = A[2][l] = A[3][l] = 0.0;
#pragma omp simd
        for (int l = 0; l < FPVS_TILE; ++l) {
            /* components.py:185-189 */
            const double rx = clipd(-A[0][l] * maxr, -maxr, maxr) * rtr + qx[l] * (1 - rtr);
            const double ry = clipd(-A[1][l] * maxr, -maxr, maxr) * rtr + qy[l] * (1 - rtr);
            const double rz = clipd(-A[2][l] * maxr, -maxr, maxr) * rtr + qz[l] * (1 - rtr);
            qx[l] = rx; qy[l] = ry; qz[l] = rz;
            /* components.py:136, :192-194 */
            const double x = 100 * (A[3][l] + 1) / 2;
            const double poly = ((c0 * x + c1) * x + c2) * x + c3;
            const double T = poly * ttr + pt[l] * (1 - ttr);
            pt[l] = T;
            const double R0 = r0[l], R1 = r1[l], R2 = r2[l], R3 = r3[l], R4 = r4[l], R5 = r5[l], R6 = r6[l], R7 = r7[l], R8 = r8[l];
            const double tx = R2 * T, ty = R5 * T, tz = R8 * T;                      /* kinematics.py:48-49 */
            /* kinematics.py:33-38 */
            const double sx = vx[l] + wx, sy = vy[l] + wy, sz = vz[l] + wz;
            const double speed = sqrt(sx * sx + sy * sy + sz * sz);
            const double f0 = k0 * (R0 * sx + R3 * sy + R6 * sz) * speed;
            const double f1 = k1 * (R1 * sx + R4 * sy + R7 * sz) * speed;
            const double f2 = k2 * (R2 * sx + R5 * sy + R8 * sz) * speed;
            const double dx = R0 * f0 + R1 * f1 + R2 * f2, dy = R3 * f0 + R4 * f1 + R5 * f2, dz = R6 * f0 + R7 * f1 + R8 * f2;
            /* components.py:235: motor heights in the world (only z matters for Ground and the done flag) */
            double collz = 0.0, any_below = 0.0;
            const double vn = vx[l] * 0 + vy[l] * 0 + vz[l] * 1;
#define FPVS_MOTOR(k) { const double mz = pz[l] + (m##k##x * R6 + m##k##y * R7 + 0.0 * R8); any_below = mz < 0.0 ? 1.0 : any_below; \
                        const double d = mz - mrad; collz += d < 0 ? (-ksp * d - kdm * vn) * 1.0 : 0.0; }   /* kinematics.py:56-59 */
            FPVS_MOTOR(0) FPVS_MOTOR(1) FPVS_MOTOR(2) FPVS_MOTOR(3)
            collz = any_below == 0.0 ? collz * groundf : 0.0;                        /* a crash returns the forces summed so far: zero (quirk Q5) */
            DN[l] = any_below;                                                        /* components.py:239-240 */
            /* components.py:242-243 */
            const double ax = (tx + 0.0 + dx + 0.0) / mass, ay = (ty + 0.0 + dy + 0.0) / mass, az = (tz + gz + dz + collz) / mass;
            /* kinematics.py:21-23 */
            px[l] += vx[l] * dt; py[l] += vy[l] * dt; pz[l] += vz[l] * dt;
            vx[l] += ax * dt; vy[l] += ay * dt; vz[l] += az * dt;
            /* helper_functions.py:19-44: E = Rz @ Ry @ Rx for (roll, pitch, yaw) = rates * dt in radians */
            const double a = rx * DEG2RAD * dt, b = ry * DEG2RAD * dt, c = rz * DEG2RAD * dt;
            const double cr = fpvs_cos(a), sr = fpvs_sin(a), cp = fpvs_cos(b), sp = fpvs_sin(b), cy = fpvs_cos(c), sy_ = fpvs_sin(c);
            /* zy = Rz @ Ry, E = zy @ Rx - written out with the zero / one terms of the oracle's 3x3 products kept */
            const double z0 = cy * cp + -sy_ * 0 + 0 * -sp, z1 = cy * 0 + -sy_ * 1 + 0 * 0, z2 = cy * sp + -sy_ * 0 + 0 * cp;
            const double z3 = sy_ * cp + cy * 0 + 0 * -sp, z4 = sy_ * 0 + cy * 1 + 0 * 0, z5 = sy_ * sp + cy * 0 + 0 * cp;
            const double z6 = 0 * cp + 0 * 0 + 1 * -sp, z7 = 0 * 0 + 0 * 1 + 1 * 0, z8 = 0 * sp + 0 * 0 + 1 * cp;
            const double E0 = z0 * 1 + z1 * 0 + z2 * 0, E1 = z0 * 0 + z1 * cr + z2 * sr, E2 = z0 * 0 + z1 * -sr + z2 * cr;
            const double E3 = z3 * 1 + z4 * 0 + z5 * 0, E4 = z3 * 0 + z4 * cr + z5 * sr, E5 = z3 * 0 + z4 * -sr + z5 * cr;
            const double E6 = z6 * 1 + z7 * 0 + z8 * 0, E7 = z6 * 0 + z7 * cr + z8 * sr, E8 = z6 * 0 + z7 * -sr + z8 * cr;
            /* kinematics.py:27-30 twice (components.py:218): Rn[i][j] = sum_k E[j][k] R[i][k] */
            double N0 = E0 * R0 + E1 * R1 + E2 * R2, N1 = E3 * R0 + E4 * R1 + E5 * R2, N2 = E6 * R0 + E7 * R1 + E8 * R2;
            double N3 = E0 * R3 + E1 * R4 + E2 * R5, N4 = E3 * R3 + E4 * R4 + E5 * R5, N5 = E6 * R3 + E7 * R4 + E8 * R5;
            double N6 = E0 * R6 + E1 * R7 + E2 * R8, N7 = E3 * R6 + E4 * R7 + E5 * R8, N8 = E6 * R6 + E7 * R7 + E8 * R8;
            const double M0 = E0 * N0 + E1 * N1 + E2 * N2, M1 = E3 * N0 + E4 * N1 + E5 * N2, M2 = E6 * N0 + E7 * N1 + E8 * N2;
            const double M3 = E0 * N3 + E1 * N4 + E2 * N5, M4 = E3 * N3 + E4 * N4 + E5 * N5, M5 = E6 * N3 + E7 * N4 + E8 * N5;
            const double M6 = E0 * N6 + E1 * N7 + E2 * N8, M7 = E3 * N6 + E4 * N7 + E5 * N8, M8 = E6 * N6 + E7 * N7 + E8 * N8;
            r0[l] = M0; r1[l] = M1; r2[l] = M2; r3[l] = M3; r4[l] = M4; r5[l] = M5; r6[l] = M6; r7[l] = M7; r8[l] = M8;
            /* components.py:248 */
            AX[l] = M0 * ax + M1 * ay + M2 * az; AY[l] = M3 * ax + M4 * ay + M5 * az; AZ[l] = M6 * ax + M7 * ay + M8 * az;
        }
    }
    for (int l = 0; l < m; ++l) {
        for (int k = 0; k < FPVO_DRONE_STATE; ++k) state[(int64_t)l * FPVO_DRONE_STATE + k] = S[k][l];
        if (accel) { accel[3 * l + 0] = AX[l]; accel[3 * l + 1] = AY[l]; accel[3 * l + 2] = AZ[l]; }
        if (done) done[l] = DN[l] != 0.0;
    }
}

/* Same contract as fpvo_drone_step_batch (fpv_oracle.h).  Returns 0, or -1 when the parameters ask for what this
 * vector form does not cover (a general object list). */
int fpvs_drone_step_batch(const fpvo_params* P, int64_t n, int steps, double* state, const double* actions, int action_per_step,
                          const double wind[3], double* accel, uint8_t* done, int threads)
{
    if (P->n_objects > 0) return -1;
    if (steps <= 0) return 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
    const int64_t tiles = (n + FPVS_TILE - 1) / FPVS_TILE;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t b = 0; b < tiles; ++b) {
        const int64_t i0 = b * FPVS_TILE;
        const int m = (int)((i0 + FPVS_TILE < n ? i0 + FPVS_TILE : n) - i0);
        tile_run(P, m, steps, state + i0 * FPVO_DRONE_STATE, actions + i0 * 4, action_per_step ? n * 4 : 0, wind,
                 accel ? accel + 3 * i0 : 0, done ? done + i0 : 0);
    }
    return 0;
}
