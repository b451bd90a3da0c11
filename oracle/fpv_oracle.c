/*
 * fpv_oracle.c - CPU restatement (float64, plain C) of the reference's per-drone step.
 *
 * TEST INFRASTRUCTURE ONLY - see fpv_oracle.h.  Parity status: PINNED by tests/golden (npz files),
 * which oracle/gen_golden.py produced by running the reference's own Drone.step / Racer.step.
 *
 * The attitude is kept as a 3x3 matrix and advanced exactly as the reference does it (including
 * the second application per step); nothing here is shared with the HIP kernel, which integrates
 * a quaternion in fp32.  Every function names the reference lines it follows (paths relative to
 * /root/reference).
 */
#include "fpv_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DEG2RAD (M_PI / 180.0)   /* np.deg2rad(x) == x * (pi/180) */

/* C = A @ B, 3x3 row-major */
static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

/* src/utils/helper_functions.py:19-36 (rotation_matrix) and :39-44
 * (euler_angles_to_rotation_matrix = R_z @ R_y @ R_x). */
void fpvo_euler_zyx_matrix(double roll, double pitch, double yaw, double E[9])
{
    const double cr = cos(roll), sr = sin(roll);
    const double cp = cos(pitch), sp = sin(pitch);
    const double cy = cos(yaw), sy = sin(yaw);
    const double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
    const double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
    const double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
    double zy[9];
    mat3_mul(Rz, Ry, zy);
    mat3_mul(zy, Rx, E);
}

/* src/utils/kinematics.py:27-30 (rotate_body_by_rates): R <- (E @ R.T).T, rates in deg/s. */
static void rotate_body_by_rates(double* R, const double rates[3], double dt)
{
    double E[9], Rn[9];
    fpvo_euler_zyx_matrix(rates[0] * DEG2RAD * dt, rates[1] * DEG2RAD * dt, rates[2] * DEG2RAD * dt, E);
    for (int i = 0; i < 3; ++i)          /* ((E @ R.T).T)[i][j] = sum_k E[j][k] * R[i][k] */
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = E[3 * j + 0] * R[3 * i + 0] + E[3 * j + 1] * R[3 * i + 1] + E[3 * j + 2] * R[3 * i + 2];
    memcpy(R, Rn, sizeof(Rn));
}

static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* src/utils/components.py:220-248 (Drone.step).  rotation_override (row-major 3x3, or NULL) and thrust_force
 * are the guidance arguments rotation_matrix= / thrust_force= (:230-232). */
void fpvo_drone_step_guided(const fpvo_params* P, double* s, const double action[4], const double wind[3],
                            const double* rotation_override, double thrust_force, double accel_out[3], uint8_t* done_out)
{
    double* p = s;
    double* v = s + 3;
    double* R = s + 6;
    double* prev_rates = s + 15;
    double* prev_thrust = s + 18;

    /* components.py:185-189: stick -> rate command, clipped, first-order low-pass */
    double rates[3];
    for (int i = 0; i < 3; ++i) {
        const double cmd = clipd(-action[i] * P->max_rates, -P->max_rates, P->max_rates);
        rates[i] = cmd * P->rates_transition_rate + prev_rates[i] * (1 - P->rates_transition_rate);
        prev_rates[i] = rates[i];
    }
    /* components.py:136,:192-194: cubic in throttle percent (np.poly1d = Horner), low-pass, no clamp */
    const double x = 100 * (action[3] + 1) / 2;
    const double* c = P->thrust_poly;
    const double poly = ((c[0] * x + c[1]) * x + c[2]) * x + c[3];
    const double T = poly * P->thrust_transition_rate + *prev_thrust * (1 - P->thrust_transition_rate);
    *prev_thrust = T;
    /* kinematics.py:48-49: thrust along the third COLUMN of the pre-update R */
    double thrust[3] = {R[2] * T, R[5] * T, R[8] * T};
    /* components.py:230-232: `if rotation_matrix is not None`: the attitude is replaced (prev_rates / prev_thrust
     * above keep the values derived from the sticks) and the thrust vector is rebuilt from thrust_force */
    if (rotation_override) {
        for (int i = 0; i < 9; ++i) R[i] = rotation_override[i];
        thrust[0] = R[2] * thrust_force; thrust[1] = R[5] * thrust_force; thrust[2] = R[8] * thrust_force;
    }

    /* kinematics.py:33-38: body-frame quadratic drag; wind is ADDED to the velocity */
    const double vs[3] = {v[0] + wind[0], v[1] + wind[1], v[2] + wind[2]};
    const double speed = sqrt(vs[0] * vs[0] + vs[1] * vs[1] + vs[2] * vs[2]);
    double fb[3], drag[3];
    for (int i = 0; i < 3; ++i) {
        const double vb = R[0 + i] * vs[0] + R[3 + i] * vs[1] + R[6 + i] * vs[2];       /* (R.T @ vs)[i] */
        fb[i] = -0.5 * P->drag_coefficients[i] * P->air_density * P->cross_section_areas[i] * vb * speed;
    }
    for (int i = 0; i < 3; ++i)
        drag[i] = R[3 * i + 0] * fb[0] + R[3 * i + 1] * fb[1] + R[3 * i + 2] * fb[2];

    /* kinematics.py:41-45 */
    const double grav[3] = {0, 0, -P->gravity * P->mass};

    /* components.py:235: motor positions in the world, pre-update pose */
    double mpos[4][3];
    for (int m = 0; m < 4; ++m)
        for (int j = 0; j < 3; ++j)
            mpos[m][j] = p[j] + (P->motor_xy[m][0] * R[3 * j + 0] + P->motor_xy[m][1] * R[3 * j + 1] + 0.0 * R[3 * j + 2]);

    /* components.py:198-214 (handle_collisions) for object_list = [Ground]: distance = z
     * (Ground.calculate_distance, :674-677), normal = [0,0,1].  The crash test `any(distances < 0)`
     * sits INSIDE the per-motor loop and returns the forces summed so far - on the first iteration,
     * i.e. zero (quirk Q5). */
    double coll[3] = {0, 0, 0};
    uint8_t done = 0;
    if (P->ground) {
        int any_below = 0;
        for (int m = 0; m < 4; ++m) any_below |= (mpos[m][2] < 0);
        if (any_below) {
            done = 1;
        } else {
            for (int m = 0; m < 4; ++m) {
                const double d = mpos[m][2] - P->motor_radius;
                if (d < 0) {   /* kinematics.py:56-59 spring_force(d, normal, velocity, k, c) */
                    const double vn = v[0] * 0 + v[1] * 0 + v[2] * 1;
                    coll[2] += (-P->ground_spring * d - P->ground_damping * vn) * 1.0;
                }
            }
        }
    }
    /* general object_list: objects in list order; per object distances of all four motors first
     * (components.py:203-204), then the per-motor loop whose first iteration returns on a crash,
     * keeping the forces of EARLIER objects (:205-212).
     *   Ground   distance = z, normal +z                         components.py:674-680
     *   Cylinder distance/normal as written, incl. the relative-vs-absolute z test in
     *            calculate_normal                                 components.py:710-729
     *   Target   |p - c| - radius, radial normal                  components.py:774-778 */
    if (P->n_objects > 0) {
        coll[0] = coll[1] = coll[2] = 0;
        done = 0;
        for (int o = 0; o < P->n_objects && !done; ++o) {
            const double cx = P->objects[o].x, cy = P->objects[o].y, cz = P->objects[o].z;
            const double rad = P->objects[o].radius, hgt = P->objects[o].height;
            double dist[4], nrm[4][3];
            for (int m = 0; m < 4; ++m) {
                const double* q = mpos[m];
                if (P->objects[o].type == 0) {
                    dist[m] = q[2]; nrm[m][0] = 0; nrm[m][1] = 0; nrm[m][2] = 1;
                } else if (P->objects[o].type == 1) {
                    const double d2 = sqrt((q[0] - cx) * (q[0] - cx) + (q[1] - cy) * (q[1] - cy)) - rad;
                    if (cz < q[2] && q[2] < cz + hgt) dist[m] = d2;
                    else {
                        const double dh = fmin(fabs(q[2] - cz), fabs(q[2] - (cz + hgt)));
                        dist[m] = sqrt(d2 * d2 + dh * dh);
                    }
                    const double rx = q[0] - cx, ry = q[1] - cy, rz = q[2] - cz;
                    if (cz < rz && rz < cz + hgt) {
                        const double nn = sqrt(rx * rx + ry * ry);
                        nrm[m][0] = rx / nn; nrm[m][1] = ry / nn; nrm[m][2] = 0;
                    } else {
                        nrm[m][0] = 0; nrm[m][1] = 0;
                        nrm[m][2] = (fabs(rz - cz) < fabs(rz - (cz + hgt))) ? -1 : 1;
                    }
                } else {
                    const double rx = q[0] - cx, ry = q[1] - cy, rz = q[2] - cz;
                    const double nn = sqrt(rx * rx + ry * ry + rz * rz);
                    dist[m] = nn - rad;
                    nrm[m][0] = rx / nn; nrm[m][1] = ry / nn; nrm[m][2] = rz / nn;
                }
            }
            if (dist[0] < 0 || dist[1] < 0 || dist[2] < 0 || dist[3] < 0) { done = 1; break; }
            for (int m = 0; m < 4; ++m) {
                const double d = dist[m] - P->motor_radius;
                if (d < 0) {
                    const double vn = v[0] * nrm[m][0] + v[1] * nrm[m][1] + v[2] * nrm[m][2];
                    const double f = -P->ground_spring * d - P->ground_damping * vn;
                    for (int j = 0; j < 3; ++j) coll[j] += f * nrm[m][j];
                }
            }
        }
    }
    /* components.py:239-240: any motor below z = 0, evaluated on the PRE-update pose, not latched */
    for (int m = 0; m < 4; ++m)
        if (mpos[m][2] < 0.0) done = 1;

    /* components.py:242-243 */
    double acc[3];
    for (int i = 0; i < 3; ++i)
        acc[i] = (thrust[i] + grav[i] + drag[i] + coll[i]) / P->mass;

    /* kinematics.py:21-23: p with the OLD v, then v, then one attitude increment ... */
    for (int i = 0; i < 3; ++i) p[i] += v[i] * P->dt;
    for (int i = 0; i < 3; ++i) v[i] += acc[i] * P->dt;
    rotate_body_by_rates(R, rates, P->dt);
    /* ... components.py:218: and the same increment AGAIN */
    rotate_body_by_rates(R, rates, P->dt);

    /* components.py:248: third return value is R_new @ acc */
    if (accel_out)
        for (int i = 0; i < 3; ++i)
            accel_out[i] = R[3 * i + 0] * acc[0] + R[3 * i + 1] * acc[1] + R[3 * i + 2] * acc[2];
    if (done_out) *done_out = done;
}

void fpvo_drone_step(const fpvo_params* P, double* s, const double action[4], const double wind[3],
                     double accel_out[3], uint8_t* done_out)
{
    fpvo_drone_step_guided(P, s, action, wind, 0, 0.0, accel_out, done_out);
}

/* Drones are independent, so the batch is cut into contiguous tiles of FPVO_TILE drones; each thread
 * takes whole tiles and walks TIME-OUTER inside a tile: per step it reads one contiguous 8 KB slice of
 * that step's action batch, and the tile's state (39 KB) stays in its L1/L2 for all steps.  (Walking
 * drone-outer / time-inner instead strides through the [steps][n][4] action array and misses the
 * cache once per step per drone.) */
#define FPVO_TILE 256

void fpvo_drone_step_batch(const fpvo_params* P, int64_t n, int steps, double* state,
                           const double* actions, int action_per_step, const double wind[3],
                           double* accel, uint8_t* done, int threads)
{
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
    const int64_t tiles = (n + FPVO_TILE - 1) / FPVO_TILE;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t b = 0; b < tiles; ++b) {
        const int64_t i0 = b * FPVO_TILE, i1 = (i0 + FPVO_TILE < n) ? i0 + FPVO_TILE : n;
        for (int t = 0; t < steps; ++t) {
            const double* at = actions + (action_per_step ? (int64_t)t * n : 0) * 4;
            const int last = t == steps - 1;
            for (int64_t i = i0; i < i1; ++i) {
                double a3[3];
                uint8_t d = 0;
                fpvo_drone_step(P, state + i * FPVO_DRONE_STATE, at + i * 4, wind, a3, &d);
                if (last) {
                    if (accel) memcpy(accel + 3 * i, a3, sizeof(a3));
                    if (done) done[i] = d;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Racer: tests/racer_drone_test.py.  Orientation lives in a scipy Rotation, i.e. a unit
 * quaternion (x,y,z,w); each step goes quaternion -> matrix -> product -> from_matrix.
 * ------------------------------------------------------------------------------------------- */

/* scipy Rotation.as_matrix for q = (x,y,z,w) */
static void quat_xyzw_to_matrix(const double q[4], double M[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[3] = 2 * (xy + zw);      M[6] = 2 * (xz - yw);
    M[1] = 2 * (xy - zw);     M[4] = -x2 + y2 - z2 + w2; M[7] = 2 * (yz + xw);
    M[2] = 2 * (xz + yw);     M[5] = 2 * (yz - xw);      M[8] = -x2 - y2 + z2 + w2;
}

/* scipy Rotation.from_matrix: largest-of-(diag, trace) branch, then normalise */
static void matrix_to_quat_xyzw(const double M[9], double q[4])
{
    const double tr = M[0] + M[4] + M[8];
    const double dec[4] = {M[0], M[4], M[8], tr};
    int choice = 0;
    for (int i = 1; i < 4; ++i)
        if (dec[i] > dec[choice]) choice = i;
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - tr + 2 * M[3 * i + i];
        q[j] = M[3 * j + i] + M[3 * i + j];
        q[k] = M[3 * k + i] + M[3 * i + k];
        q[3] = M[3 * k + j] - M[3 * j + k];
    } else {
        q[0] = M[7] - M[5];
        q[1] = M[2] - M[6];
        q[2] = M[3] - M[1];
        q[3] = 1 + tr;
    }
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= nrm;
}

/* src/utils/components.py:43-54 (PID.__call__), np.clip(x, lo, hi) = min(max(x, lo), hi) */
double fpvo_pid_call(const double k[8], double st[4], double current, double target)
{
    const double kP = k[0], kI = k[1], kD = k[2], dt = k[3], iclip = k[4], omin = k[5], omax = k[6], dtr = k[7];
    const double error = current - target;                                                  /* :44 */
    st[0] = clipd(0.99 * st[0] + error * dt, -iclip, iclip);                                /* :46 */
    double derivative = clipd((1 - (st[3] != 0.0)) * (error - st[2]) / dt, -1, 1);          /* :48 */
    derivative = (1 - dtr) * st[1] + dtr * derivative;                                      /* :49 */
    st[1] = derivative;                                                                     /* :50 */
    st[3] = 0.0;                                                                            /* :52 */
    st[2] = error;                                                                          /* :53 */
    return clipd(kP * error + kI * st[0] + kD * derivative, omin, omax);                    /* :54 */
}

/* tests/racer_drone_test.py:95-103 (Racer.step) with PID.step of :22-32 */
void fpvo_racer_step(const fpvo_params* P, double* s, const double action[4])
{
    double* p = s;
    double* v = s + 3;
    double* q = s + 6;          /* x,y,z,w */
    double* w = s + 10;
    double* ierr = s + 13;
    double* lerr = s + 16;
    double* first = s + 19;
    const double dt = P->dt;

    double* dflt = s + 20;
    double torque[3];
    if (P->racer_pid_variant == 1) {                    /* components.PID.__call__(current = omega, target = action) */
        for (int i = 0; i < 3; ++i) {
            const double k[8] = {P->racer_pid[i][0], P->racer_pid[i][1], P->racer_pid[i][2], dt, P->pid_integral_clip,
                                 P->pid_min_output, P->pid_max_output, P->pid_derivative_transition_rate};
            double st[4] = {ierr[i], dflt[i], lerr[i], *first};
            torque[i] = fpvo_pid_call(k, st, w[i], action[i]);
            ierr[i] = st[0]; dflt[i] = st[1]; lerr[i] = st[2];
        }
    } else
    for (int i = 0; i < 3; ++i) {                       /* :22-32 */
        const double err = action[i] - w[i];
        ierr[i] += err * dt;
        double derr = (err - lerr[i]) / dt;
        if (*first != 0.0) derr = 0;
        lerr[i] = err;
        torque[i] = P->racer_pid[i][0] * err + P->racer_pid[i][1] * ierr[i] + P->racer_pid[i][2] * derr;
    }
    *first = 0.0;
    for (int i = 0; i < 3; ++i)                         /* :98 */
        w[i] = 1 * w[i] + torque[i] * dt / P->racer_inertia[i];

    /* :99 orientation <- orientation @ R.from_euler("XYZ", omega)  (intrinsic: Rx @ Ry @ Rz);
     * omega is used as an angle as written; racer_omega_dt scales it by dt instead. */
    const double k = P->racer_omega_dt ? dt : 1.0;
    const double a = w[0] * k, b = w[1] * k, c = w[2] * k;
    const double Rx[9] = {1, 0, 0, 0, cos(a), -sin(a), 0, sin(a), cos(a)};
    const double Ry[9] = {cos(b), 0, sin(b), 0, 1, 0, -sin(b), 0, cos(b)};
    const double Rz[9] = {cos(c), -sin(c), 0, sin(c), cos(c), 0, 0, 0, 1};
    double xy[9], inc[9], M[9], Mn[9];
    mat3_mul(Rx, Ry, xy);
    mat3_mul(xy, Rz, inc);
    quat_xyzw_to_matrix(q, M);
    mat3_mul(M, inc, Mn);
    matrix_to_quat_xyzw(Mn, q);
    quat_xyzw_to_matrix(q, M);

    for (int i = 0; i < 3; ++i) {                       /* :100-103 */
        const double force = action[3] * M[3 * i + 2];
        const double acc = force / P->racer_mass;
        v[i] = P->racer_velocity_damping * v[i] + acc * dt;
        p[i] += v[i] * dt;
    }
}

void fpvo_racer_step_batch(const fpvo_params* P, int64_t n, int steps, double* state,
                           const double* actions, int action_per_step, int threads)
{
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
    const int64_t tiles = (n + FPVO_TILE - 1) / FPVO_TILE;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t b = 0; b < tiles; ++b) {
        const int64_t i0 = b * FPVO_TILE, i1 = (i0 + FPVO_TILE < n) ? i0 + FPVO_TILE : n;
        for (int t = 0; t < steps; ++t) {
            const double* at = actions + (action_per_step ? (int64_t)t * n : 0) * 4;
            for (int64_t i = i0; i < i1; ++i)
                fpvo_racer_step(P, state + i * FPVO_RACER_STATE, at + i * 4);
        }
    }
}

/* ---- comparison helpers ---- */

/* src/utils/helper_functions.py:100-117 */
void fpvo_quat_wxyz_to_matrix(const double q[4], double R[9])
{
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    R[0] = 1 - 2 * qy * qy - 2 * qz * qz; R[1] = 2 * qx * qy - 2 * qz * qw;     R[2] = 2 * qx * qz + 2 * qy * qw;
    R[3] = 2 * qx * qy + 2 * qz * qw;     R[4] = 1 - 2 * qx * qx - 2 * qz * qz; R[5] = 2 * qy * qz - 2 * qx * qw;
    R[6] = 2 * qx * qz - 2 * qy * qw;     R[7] = 2 * qy * qz + 2 * qx * qw;     R[8] = 1 - 2 * qx * qx - 2 * qy * qy;
}

/* The reference's own converter (helper_functions.py:65-80) divides by 4*qw and breaks as the
 * trace approaches -1; comparisons use this branch-on-largest form instead (w >= 0 afterwards). */
void fpvo_matrix_to_quat_wxyz(const double R[9], double q[4])
{
    double x[4];
    matrix_to_quat_xyzw(R, x);
    const double sgn = x[3] < 0 ? -1.0 : 1.0;
    q[0] = sgn * x[3]; q[1] = sgn * x[0]; q[2] = sgn * x[1]; q[3] = sgn * x[2];
}

int fpvo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
