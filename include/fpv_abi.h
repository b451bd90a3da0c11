/*
 * fpv_abi.h - C ABI of the MI355X batched FPV drone stepper (libfpv_hip.so).
 *
 * The reference has no FFI for this path: its boundary is the Python method pair
 *   Drone.reset(position, velocity, ypr)                 /root/reference/src/utils/components.py:150-169
 *   Drone.step(action, wind_velocity_vector, object_list) /root/reference/src/utils/components.py:220-248
 * (plus Racer.reset/step, /root/reference/tests/racer_drone_test.py:85-103).  The entry points
 * below are what a ctypes binding of that pair calls for N drones at once; each one names the
 * reference lines it replaces.  Plain pointers and sizes only - no torch, no C++ types.
 *
 * Conventions
 *   - every function returns FPV_OK (0) or a negative FPV_E* code; fpv_last_error() gives the
 *     message of the calling thread's last failure.  Nothing throws across the boundary.
 *   - device buffers are owned by the caller (the Python host allocates them as torch tensors);
 *     the library never allocates, frees or copies device state.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue work; asynchronous
 *     kernel faults surface at the caller's next synchronisation.
 *   - a handle is bound to one device and is not thread-safe; use one per GPU / host thread.  Every call makes the
 *     handle's device current for its own launches and restores the caller's current device before returning.
 */
#ifndef FPV_ABI_H
#define FPV_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: 64-bit step counter (fpv_set_step_counter takes uint64_t, fpv_get_step_counter added; the stick-noise stream is
 *    unchanged below 2^32 steps and no longer repeats beyond); fpv_set_tuning removed (2 / 4 drones per lane and
 *    256-thread workgroups lost every measurement); fpv_comm_info added; fpv_step_n reads action rows only. */
/* 5: fpv_diag_busy added (a time-bounded one-wave kernel: lets a host tell whether two streams sit on different
 *    hardware queues - the split-phase API picks its partition streams that way); the in-kernel stick-noise generator
 *    rebuilt for cost (Philox4x32-7, four normals per call from a table-driven inverse CDF - no logarithm): streams differ from ABI <= 4 (the reference's profile is
 *    unseeded, /root/reference/tests/noise_smooth_test.py:6-12: there never was a stream to stay compatible with). */
/* 6: fpv_buffers_t.action_f16 / reserved0 (binary16 stick rows, added late in ABI 5) removed: measured at no time gain
 *    for ten more kernels - a half-precision policy casts its sticks (`.float()`); sizeof(fpv_buffers_t) shrinks by 8.
 *    fp16 state: the stored quaternion fields saturate at +-16383 instead of wrapping (unit quaternions: unchanged bits). */
/* 7: fpv_set_rotation / fpv_get_rotation added: the fp32 drone step kernels walk the population from a start block that moves
 *    back by a cache's worth of drones per launch - the L2s' when the state overflows them (2^20 drones: 11 % less time), the
 *    Infinity Cache's beyond that (2^23 drones: up to 25 % less) - same results; automatic by default. */
/* 8: the cache model behind the rotation and the row stride is CHECKED against the device (hipGetDeviceProperties at fpv_create):
 *    fpv_check_cache_model / fpv_device_cache_model / fpv_get_cache_model / fpv_recommended_ld_device added.  A device that is not
 *    gfx950 with 256 compute units and a 4 MiB L2 per XCD (e.g. a CPX compute partition) gets the plain order and the
 *    conservative stride - same results.  Every single-step kernel rotates (drone fp32 / fp16 state / AoS head / Racer), as ABI 7
 *    already did; its comment said "fp32 drone" only.  fpv_encoding_id added (what a checkpoint's fp16 words / noise stream mean). */
#define FPV_ABI_VERSION 8

enum {
    FPV_OK = 0,
    FPV_EINVAL = -1,  /* bad argument (null pointer, n <= 0, bad mode, struct_size mismatch) */
    FPV_EHIP = -2,    /* a HIP runtime call failed; message carries hipGetErrorString */
    FPV_ENODEV = -3,  /* no usable GPU / device index out of range */
    FPV_EALIGN = -4,  /* buffer alignment or leading dimension violates the layout rules */
    FPV_EPARAM = -5   /* physically meaningless parameter (dt <= 0, mass <= 0, ...) */
};

enum { FPV_MODE_DRONE = 0, FPV_MODE_RACER = 1 };

/* Rows of the SoA state matrix state[rows][ld] (fp32).  Row r of drone i is state[r*ld + i].
 * FPV_MODE_DRONE: the mutable state of Drone (components.py:151-169) with the attitude held as a
 * unit quaternion (w,x,y,z), body->world, instead of the 3x3 matrix. */
enum {
    FPV_PX = 0, FPV_PY, FPV_PZ,          /* state[0:3]   position, m                     */
    FPV_VX, FPV_VY, FPV_VZ,              /* state[3:6]   velocity, m/s                   */
    FPV_QW, FPV_QX, FPV_QY, FPV_QZ,      /* rotation_matrix as quaternion                */
    FPV_RX, FPV_RY, FPV_RZ,              /* prev_rates, deg/s (components.py:189)        */
    FPV_THRUST,                          /* prev_thrust, N    (components.py:194)        */
    FPV_DRONE_ROWS                       /* = 14 */
};
/* FPV_MODE_RACER: Racer state (racer_drone_test.py:70-83) + its three PID integrators (:13-20). */
enum {
    FPV_R_OMEGA = 10,                    /* rows 10..12 angular_velocity                 */
    FPV_R_IERR = 13,                     /* rows 13..15 PID i_error                      */
    FPV_R_LERR = 16,                     /* rows 16..18 PID last_error                   */
    FPV_R_FIRST = 19,                    /* 1.0 until the first PID step                 */
    FPV_R_OMEGA_LO = 20,                 /* rows 20..22 low words of angular_velocity: Racer.step AS WRITTEN turns by
                                            omega RADIANS per step (racer_drone_test.py:99), so omega is carried as an
                                            fp32 (hi, lo) pair; unused (never read or written) with racer_omega_dt = 1 */
    FPV_R_IERR_LO = 23,                  /* rows 23..25 low words of the PID i_error, same reason                */
    FPV_R_DFILT = 26,                    /* rows 26..28 prev_derivative of components.PID (components.py:50); only
                                            touched with racer_pid_variant = 1                                   */
    FPV_RACER_ROWS = 29
};

enum {
    FPV_FLAG_AUTO_RESET = 1u,   /* re-initialise a lane in-kernel when it reports done */
    FPV_FLAG_GROUND = 2u,       /* ground plane z = 0 in object_list: per-motor spring contact,
                                   Drone.handle_collisions with a Ground object (components.py:198-214) */
    FPV_FLAG_STICK_NOISE = 8u,  /* drone mode, fp32 state: the kernel advances an EMA-smoothed Gaussian stick noise
                                   per drone and channel (the profile of tests/noise_smooth_test.py:6-12, Philox4x32-7 + a table-driven inverse normal CDF,
                                   keyed by seed / global drone id / step) and ADDS gain * noise to the action
                                   (clipped to [-1,1]); fpv_buffers_t.action may then be NULL (pure noise sticks) */
    FPV_FLAG_FP16_STATE = 4u    /* drone mode only: v, q, prev_rates, prev_thrust stored as eleven 16-bit words per drone in
                                   fpv_buffers_t.state_h; fpv_buffers_t.state holds only the 3 position rows in
                                   fp32; arithmetic stays fp32 (BASELINE config 4) */
};
/* state_h under FPV_FLAG_FP16_STATE: FPV_HALF_PAIR_ROWS rows of ld 32-bit word pairs (low 16 bits first):
 * (vx,vy) (vz,v_low) (qa,qb) (qc,rx) (ry,rz), followed by ONE row of ld single halves holding prev_thrust:
 * (FPV_HALF_PAIR_ROWS * 2 + 1) * ld 16-bit words in all, 22 bytes per drone.  Encoding (ABI 5; csrc/fpv_math.h,
 * fpv_pack_half / fpv_unpack_half; fpv_widen_state decodes a whole batch):
 *   vx vy vz rx ry rz thrust   IEEE binary16 (v: round toward zero of the stochastically rounded value; rates / thrust: nearest even)
 *   v_low                      bits 0-4 / 5-9 / 10-14: the next five mantissa bits of vx / vy / vz (v has 15 mantissa bits in all)
 *   qa qb qc                   "smallest three": bits 0-14 = 15-bit two's-complement fixed point, value / 23168, of the three
 *                              quaternion components that are NOT the largest in magnitude, in w x y z order; bit 15 of qa and
 *                              of qb = bits 0 and 1 of the index of the dropped component, which is positive and equals
 *                              sqrt(1 - qa^2 - qb^2 - qc^2); stochastically rounded.
 * The kernel never issues a 2-byte access: the two lanes of an even/odd drone pair share the dword of the thrust row and
 * exchange their halves in registers. */
#define FPV_HALF_PAIR_ROWS 5
#define FPV_HALF_ROWS_TOTAL_HALVES 11    /* halves per drone in state_h */
#define FPV_OBS_AOS_DIM 16

/* Host-side description of one drone type; doubles, narrowed to fp32 by fpv_create.
 * Field sources: components.py:92-100 (dt, gravity, mass, drag, areas), :120-125 (motor_xy),
 * :134-136 (thrust_poly), :185-194 (max_rates, transition rates), kinematics.py:33 (air_density),
 * racer_drone_test.py:8,:70-83,:102 (racer_*). */
typedef struct fpv_params {
    uint32_t struct_size;             /* = sizeof(fpv_params_t) */
    uint32_t mode;                    /* FPV_MODE_* */
    uint32_t flags;                   /* FPV_FLAG_* */
    uint32_t racer_omega_dt;          /* 0: rotate by omega per step as the reference writes it; 1: omega*dt */
    double dt;
    double gravity;
    double mass;                      /* kg */
    double max_rates;                 /* deg/s */
    double rates_transition_rate;
    double thrust_transition_rate;
    double thrust_poly[4];            /* c3,c2,c1,c0 of thrust[N] over throttle percent */
    double drag_coefficients[3];
    double cross_section_areas[3];    /* m^2 */
    double air_density;
    double motor_xy[4][2];            /* body-frame motor positions (z = 0), m */
    double init_position[3];          /* used by fpv_reset defaults and FPV_FLAG_AUTO_RESET */
    double init_velocity[3];
    double init_quat[4];              /* w,x,y,z */
    double ceiling;                   /* auto-reset when |z| > ceiling; +inf disables */
    double goal[3];                   /* reward = -|p - goal| (build-defined; the reference has none) */
    double racer_mass;
    double racer_inertia[3];
    double racer_pid[3][3];           /* [axis][kP,kI,kD] */
    double racer_velocity_damping;
    double motor_radius;              /* contact starts at distance < motor_radius (components.py:121), m */
    double ground_spring;             /* N/m   (handle_collisions default 100, components.py:198) */
    double ground_damping;            /* N s/m (handle_collisions default 0) */
    double noise_transition;          /* FPV_FLAG_STICK_NOISE: x_s <- (1-tau) x_s + tau N(0,1); noise_smooth_test.py:5 uses 0.1 */
    double noise_gain;                /* sticks += noise_gain * x_s */
    uint64_t noise_seed;              /* Philox key */
    uint64_t drone_id_offset;         /* global id of this handle's drone 0 (shard offset): streams are keyed by global id */
    /* FPV_MODE_RACER rate loop semantics.  0: PID.step of tests/racer_drone_test.py:22-32 (error = desired - actual,
     * plain integral, raw derivative); 1: PID.__call__ of src/utils/components.py:43-54 (error = current - target,
     * integral <- clip(0.99*integral + error*dt, +-integral_clip), derivative clipped to +-1 then low-passed with
     * derivative_transition_rate, output clipped to [min_output, max_output]); gains come from racer_pid either way */
    uint32_t racer_pid_variant;
    uint32_t _reserved0;
    double pid_integral_clip;         /* components.py:16 defaults: 1 */
    double pid_min_output;            /*                            0.3 */
    double pid_max_output;            /*                            1 */
    double pid_derivative_transition_rate;   /*                     0.5 */
} fpv_params_t;

/* Analytic collision objects = the reference's object_list (components.py:198-214) in list order.
 * Ground: plane z = 0 (components.py:646-680); Cylinder: axis along +z from (x,y,z), radius, height
 * (:685-729); Sphere: a Target (:753-778) - update x,y,z before each step for a moving target
 * (simulator.py:87).  Gates and the Trail never collide in the reference (:202) and have no entry. */
#ifndef FPV_MAX_OBJECTS
#define FPV_MAX_OBJECTS 8
#endif
enum { FPV_OBJ_GROUND = 0, FPV_OBJ_CYLINDER = 1, FPV_OBJ_SPHERE = 2 };
typedef struct fpv_object { int32_t type; float x, y, z, radius, height; } fpv_object_t;
typedef struct fpv_objects { int32_t count; fpv_object_t obj[FPV_MAX_OBJECTS]; } fpv_objects_t;

/* Device buffers of one batch.  Only `state` is mandatory for fpv_reset; `state` and `action`
 * for fpv_step.  NULL optional pointers skip that output. */
typedef struct fpv_buffers {
    float* state;            /* [rows][ld] SoA, 16-byte aligned */
    int64_t ld;              /* row stride in floats, >= n, multiple of 4 */
    const float* action;     /* [n][4] = roll, pitch, yaw, throttle per drone (components.py:181-186), 16-byte aligned;
                                or, when action_ld > 0, SoA [4][action_ld] */
    float* reward;           /* [n] */
    uint8_t* done;           /* [n] one byte per drone, exactly 0 or 1 (Drone.done, components.py:236-240): a C99 bool /
                                numpy.bool_ / torch.bool array can be passed as it is */
    uint64_t* done_bits;     /* [ceil(n/64)] bit i%64 of word i/64 = done[i]; 8-byte aligned */
    float* accel;            /* [3][ld] R_new @ acc, the third value Drone.step returns (components.py:248) */
    float* ep_return;        /* [n] running episode return (read-modify-write) */
    int32_t* ep_length;      /* [n] running episode length */
    float* last_return;      /* [n] written when a lane reports done */
    int32_t* last_length;    /* [n] */
    float wind[3];           /* wind_velocity_vector of this step (kinematics.py:35: ADDED to v) */
    uint32_t rounding_seed;  /* FPV_FLAG_FP16_STATE: mixed with the handle's 64-bit step counter for the stochastic rounding
                                (step t of the handle rounds with rounding_seed + t, the counter's high word folded in) */
    uint16_t* state_h;       /* FPV_FLAG_FP16_STATE: [FPV_HALF_PAIR_ROWS][ld] word pairs (4 bytes each) + [ld] thrust halves
                                (encoding above), 8-byte aligned; else unused */
    float* pos_comp;         /* [6][ld] Kahan compensation of the p and v accumulations, or NULL (plain fp32 sums).
                                Keeps p, v within ~1 ulp over 10^4+ steps (config 1); +48 B per env-step; drone mode,
                                fp32 state; combines with stick noise and objects, not with obs_aos */
    float* noise_state;      /* FPV_FLAG_STICK_NOISE: [4][ld] EMA stick-noise state (read-modify-write); else unused */
    float* action_out;       /* [n][4] the action actually applied (after noise and clipping), 16-byte aligned, or NULL */
    int64_t action_ld;       /* 0: `action` is [n][4]; > 0: `action` is [4][action_ld] (row stride in floats, >= n)
                                - the layout of a `W[4,D] @ obs[D,n]` GEMM output; fp32 drone kernel */
    const struct fpv_objects* objects; /* HOST pointer, read during the call: the step's object_list, or NULL.
                                Drone mode, fp32 state; combines with stick noise and pos_comp; not with obs_aos or
                                FPV_FLAG_GROUND (put a Ground entry in the list instead) */
    float* obs_aos;          /* [n][FPV_OBS_AOS_DIM] row-major observation per drone, 16-byte aligned, or NULL:
                                p3, v3, q4 (wxyz), prev_rates3, R_new@acc 3 - the values Drone.step returns
                                (components.py:247-248) gathered in one row; drone mode, fp32 state only */
    int64_t done_bits_stride;/* fpv_rollout / fpv_step_n: step t writes its bit mask at done_bits + t*done_bits_stride
                                words (>= ceil(n/64)); 0 = every step overwrites the same mask */
    const float* rotation_override; /* [n][9] row-major body->world rotation matrices, or NULL: the guidance call shape
                                Drone.step(..., rotation_matrix=R, thrust_force=f) (components.py:230-232, simulator.py:110):
                                after action2force has advanced prev_rates / prev_thrust from the sticks, the attitude is
                                REPLACED by R and the thrust becomes f * R[:,2]; drag, motor positions, collisions and the
                                attitude increment of the step start from R.  fpv_step only; drone mode, fp32 state,
                                caller-supplied sticks; combines with objects / FPV_FLAG_GROUND */
    const float* thrust_override;   /* [n] thrust_force [N] of the same call; required with rotation_override.  A NaN entry
                                leaves that drone un-overridden (its own attitude and low-passed thrust) */
    uint16_t* state_h_thrust;/* FPV_FLAG_FP16_STATE: the row of prev_thrust halves when it does NOT follow the pair rows at
                                state_h + 2 * FPV_HALF_PAIR_ROWS * ld - i.e. for a handle that steps a column range [lo, hi) of a
                                larger batch (state_h moved by 2 * lo halves, this pointer = the batch's thrust row + lo halves;
                                4-byte aligned: lo even); NULL = the row follows the pair rows */
} fpv_buffers_t;

typedef struct fpv_env* fpv_handle_t;

int fpv_abi_version(void);
/* sizeof of the ABI structs as this library was compiled (0 fpv_params_t, 1 fpv_buffers_t, 2 fpv_objects_t,
 * 3 fpv_pid_params_t):
 * lets a foreign-language binding verify its struct declarations at load time */
int fpv_sizeof(int which);
/* rows of the state matrix for a mode (FPV_DRONE_ROWS / FPV_RACER_ROWS), or FPV_EINVAL */
int fpv_state_rows(int mode);
/* bytes each env-step must move at minimum (state R+W, action R, reward+done W) - roofline bookkeeping */
int fpv_algorithmic_bytes(int mode);
/* same for a live handle: FPV_FLAG_FP16_STATE moves 3*4 + 5*4 + 2 = 34 bytes of state each way = 89 B; the Racer as
 * written adds its six (hi, lo) rows (229 B), components.PID its three derivative rows (+24 B) */
int fpv_handle_algorithmic_bytes(fpv_handle_t h);

/* Replaces Drone.__init__'s physics set-up (components.py:86-142) / Racer.__init__ (:68-83).
 * Validates and narrows the parameters; binds to `device`.  No device allocation.
 * n <= 2^28 drones per handle (32-bit lane byte offsets into 16-byte action rows).
 * A handle need not own whole buffers: created with n = hi - lo and params->drone_id_offset + lo, and given every
 * fpv_buffers_t pointer moved by lo elements (same ld; done_bits by lo / 64 words; lo a multiple of 128; fp16 state:
 * state_h by 2 * lo halves and state_h_thrust = the thrust row + lo halves), it
 * steps the COLUMN RANGE [lo, hi) of a larger batch.  Several such handles on streams of their own are independent
 * kernel chains over one set of tensors - the split-phase layout (fpyv_amd.env.FpvVecEnv(partitions=P),
 * examples/c_host/main.c `split`): bit-identical to the single batch, and the chains hide part of each other's
 * per-launch floor. */
int fpv_create(const fpv_params_t* params, int64_t n, int device, fpv_handle_t* out);
void fpv_destroy(fpv_handle_t h);

/* Replaces Drone.reset (components.py:150-169) / Racer.reset (:85-93) for the lanes whose mask
 * byte is non-zero (mask NULL = all).  position/velocity/ypr_deg are [n][3] device arrays or NULL
 * (= the defaults in fpv_params_t); ypr is consumed as (roll,pitch,yaw) degrees like the reference.
 * Zeroes prev_rates, prev_thrust, PID state and episode counters of the reset lanes. */
int fpv_reset(fpv_handle_t h, const fpv_buffers_t* b, const uint8_t* mask, const float* position,
              const float* velocity, const float* ypr_deg, void* stream);

/* Replaces one Drone.step (components.py:220-248; object_list via fpv_buffers_t.objects, the guidance arguments
 * rotation_matrix= / thrust_force= via rotation_override / thrust_override) / Racer.step (:95-103) per drone. */
int fpv_step(fpv_handle_t h, const fpv_buffers_t* b, void* stream);

/* k consecutive steps, one launch each, with no host work in between: step t reads
 * actions + t*action_stride floats (action_stride = 0 holds b->action) and, when the strides are
 * non-zero, writes reward/done at + t*out_stride elements. */
int fpv_rollout(fpv_handle_t h, const fpv_buffers_t* b, int k, int64_t action_stride,
                int64_t out_stride, void* stream);

/* The same k steps as fpv_rollout - bit for bit - in ONE launch: each lane keeps its drone's state (and
 * noise / Kahan / episode accumulators) in registers for the k steps, streams step t's action from
 * action + t*action_stride while step t-1 computes, and writes reward/done (and done_bits) per step only
 * when the strides are non-zero, otherwise after the last step.  This is the open-loop / in-kernel-noise
 * loop `for i in range(time_steps): drone.step(...)` of src/core/simulator.py:83-156 without the
 * 112-byte state round trip per step: (16 + 5 + 112/k) B per env-step instead of 133 B.
 * Supported: drone mode (fp32 or fp16 state; stick noise, objects, Kahan rows in any combination) and
 * racer mode; obs_aos and SoA sticks (action_ld > 0) are refused: an observation row per step and a policy's
 * [4, n] output are closed-loop needs - use fpv_step (or fpv_rollout). */
int fpv_step_n(fpv_handle_t h, const fpv_buffers_t* b, int k, int64_t action_stride,
               int64_t out_stride, void* stream);

/* Drone.step's RETURN VALUE for every drone (components.py:247-248), after a step: rt [n][3][3] = rotation_matrix.T,
 * gyro [n][3][3] = euler_angles_to_rotation_matrix(*rates) - the low-passed rates in deg/s used as radians, as the
 * reference does -, acc [n][3] = rotation_matrix @ acceleration (copied from fpv_buffers_t.accel, which the step wrote;
 * null = not wanted).  One small kernel instead of a few dozen tensor operations on the host side of the boundary;
 * fp32 drone state only. */
int fpv_return_triple(fpv_handle_t h, const fpv_buffers_t* b, float* rt, float* gyro, float* acc, void* stream);

/* FPV_FLAG_FP16_STATE handles: the whole state as 14 fp32 rows out[14][out_ld] (same row numbering as the fp32 state) -
 * position rows copied, the eleven 16-bit words decoded exactly as the step kernel decodes them (v with its low words, q
 * rebuilt from its three stored components: a unit quaternion).  For a caller that reads the state each step (an
 * observation); one launch. */
int fpv_widen_state(fpv_handle_t h, const fpv_buffers_t* b, float* out, int64_t out_ld, void* stream);

/* The 64-bit step index that keys the stick-noise stream (Philox4x32-7 counter = global drone id, step index; key =
 * noise_seed) and the stochastic rounding counts the steps a handle has launched, from 0: set it to resume / replay a
 * run, read it to checkpoint one.  2^64 steps do not wrap in practice (2^32 took 5.5 h at the k-step kernel's rate,
 * which is why the 32-bit counter of ABI <= 3 was widened); streams below 2^32 steps are those of ABI <= 3 bit for bit.
 * A call that is refused (bad argument, failed launch) leaves the counter where it was; fpv_rollout advances it by the
 * launches that were accepted before the failing one. */
int fpv_set_step_counter(fpv_handle_t h, uint64_t step);
int fpv_get_step_counter(fpv_handle_t h, uint64_t* step);

/* Rotation of the traversal (every single-step kernel - drone fp32, fp16 state, AoS head, Racer -, launched by fpv_step /
 * fpv_rollout / fpv_rollout_graph; the k-step kernels of fpv_step_n keep the drone in registers and have nothing to find again;
 * no reference counterpart - the reference steps one drone).  Every launch of a dependent chain re-reads the state the previous launch wrote.  MI355X keeps the most
 * recently touched 256 MiB in its Infinity Cache; a population whose state is larger than that, walked in the same order every
 * launch, finds nothing of it there (cyclic access).  With rotation the launch starts `drones` BEFORE the drone at which the
 * previous launch started - i.e. on the rows the previous launch wrote last - and wraps around, ascending addresses all the
 * way; the results do not depend on the order (bit-identical).  The same holds one level up: the eight 4 MiB L2s keep the last
 * 32 MiB across a kernel boundary.  drones = -1 (default): automatic - the drones whose WRITTEN bytes (state rows, reward,
 * done and whatever else the call's buffers ask for and re-reads: Kahan rows, noise rows, episode sums ...; the accel rows and the AoS
 * observation head leave with a streaming hint and are not counted) fill 61/64 of the cache level that
 * a launch overflows (2^19 drones for the plain kernel's 61 B beyond the L2s, 2^22 beyond the Infinity Cache; whole rounds of the
 * eight XCDs), 0 when a launch writes less than the L2s hold; 0: plain order; > 0: that many drones (rounded down to whole
 * 128-drone workgroups).  fpv_get_rotation returns the value of the last launch (before the first: the estimate for reward
 * and done only).
 * Two cache tiers, one rule: which tier applies is decided per launch from what that launch writes.  A hipGraph replay
 * (fpv_rollout_graph) carries its own rotation, counted from its first node - a replay begins where the previous replay began
 * (one launch in k starts on cold rows) and neither reads nor moves the start that fpv_step / fpv_rollout keep in the handle:
 * mixing the two APIs on one handle is harmless (same results, each keeps its own order).
 * The automatic rule is a model of ONE device - gfx950 in single-partition mode: 256 compute units = eight XCDs with a 4 MiB L2
 * each behind a 256 MiB Infinity Cache, workgroups handed to the XCDs round-robin.  fpv_create asks the device
 * (hipGetDeviceProperties: architecture, compute units, L2 size); when the answer is anything else the automatic setting is the
 * plain order (fpv_get_rotation then reports 0 and leaves the reason in fpv_last_error(); fpv_get_cache_model has it too).  An
 * explicit request (drones > 0) is honoured on any device. */
int fpv_set_rotation(fpv_handle_t h, int64_t drones);
int fpv_get_rotation(fpv_handle_t h, int64_t* drones);

/* The cache model and what a device says about itself. */
typedef struct fpv_cache_model_t {
    uint32_t struct_size;           /* sizeof(fpv_cache_model_t) of the library that filled it */
    int32_t matches;                /* 1: the device is the one the model was measured on - rotation and L2-aware stride apply */
    int32_t compute_units;          /* hipDeviceProp_t.multiProcessorCount: what this process sees (a compute partition shows fewer) */
    int32_t xcds;                   /* 8 when matches (HIP does not report it: implied by gfx950 with all 256 CUs), else 0 */
    int64_t l2_bytes_per_xcd;       /* hipDeviceProp_t.l2CacheSize (0: not reported by the runtime) */
    int64_t infinity_cache_bytes;   /* 256 MiB when matches (HIP does not report it), else 0 */
    char arch[64];                  /* hipDeviceProp_t.gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    char reason[256];               /* matches == 0: what differs and what the library does instead; else "" */
} fpv_cache_model_t;
/* the rule itself, host arithmetic only (no device needed): would a device with these properties get the model? */
int fpv_check_cache_model(const char* arch, int compute_units, int64_t l2_bytes_per_xcd, fpv_cache_model_t* out);
/* the rule applied to device `device` (FPV_ENODEV without one) / what fpv_create found for this handle */
int fpv_device_cache_model(int device, fpv_cache_model_t* out);
int fpv_get_cache_model(fpv_handle_t h, fpv_cache_model_t* out);

/* Same k steps as fpv_rollout, replayed from a hipGraph cached in the handle: for small, launch-bound
 * batches (a 4096-drone step is ~2 us of kernel behind ~4 us of launch).  The graph is rebuilt only when
 * its SHAPE changes (k, strides, launch geometry, parameters, which optional buffers are present); new
 * buffer addresses alone are patched into the instantiated graph.  Frozen arguments mean no per-launch
 * step index, so handles with FPV_FLAG_STICK_NOISE or FPV_FLAG_FP16_STATE are served by the k-step kernel
 * (fpv_step_n: the same k steps bit for bit, and cheaper than a replay).
 * The same holds for a graph the CALLER captures around fpv_step / fpv_rollout (stream capture records the launches
 * with the step index they had at capture time): fine for plain handles, wrong - a repeating noise stream - for
 * FPV_FLAG_STICK_NOISE / FPV_FLAG_FP16_STATE handles, whose launches must be issued, not replayed. */
int fpv_rollout_graph(fpv_handle_t h, const fpv_buffers_t* b, int k, int64_t action_stride,
                      int64_t out_stride, void* stream);

/* Replace the drone type of a live handle (e.g. domain randomisation between episodes). */
int fpv_set_params(fpv_handle_t h, const fpv_params_t* params);

/* Row stride (in floats) to allocate for n drones.  Up to 2^18 drones: n rounded up to 64, padded so that the stride in
 * bytes is at least 1 KiB past a multiple of 8 KiB (strides at or near a multiple of 8 KiB put all 14
 * rows on the same HBM channel/bank set).  Beyond: the smallest ld >= n that is 256 mod 512 floats (1 KiB past a multiple of
 * 2 KiB: the best class of stride at every measured population), moved by multiples of 64 floats where - up to 2^21 drones - that
 * stride would make the rows of a drone block share their sets in an XCD's L2 (2^19 drones: n + 320 instead of n + 256 floats,
 * 10.7 against 13.2 us per launch).  Host arithmetic only: no device needed.  Any ld >= n that is a multiple of 4
 * is accepted by fpv_step; results do not depend on ld.
 * fpv_recommended_ld is the rule FOR THE MI355X the model was measured on, whatever device is present (or none);
 * fpv_recommended_ld_device asks `device` first and returns the first, model-free rule (64-float rounding + the 8 KiB pad) for
 * every n when the device is not that one (fpv_device_cache_model; FPV_ENODEV without a device).  Allocate with the latter. */
int64_t fpv_recommended_ld(int64_t n);
int64_t fpv_recommended_ld_device(int64_t n, int device);

/* Diagnostics only: dst[i] = src[i] for n_floats fp32 values with the step kernel's access shape
 * (one dword per lane); a known-byte-count launch for calibrating rocprofv3 byte counters. */
int fpv_diag_stream_copy(float* dst, const float* src, int64_t n_floats, void* stream);
/* the same copy with 16 bytes per lane (n_floats a multiple of 4, 16-byte aligned pointers): the streaming ceiling of
 * the chip on this box - bench.py times it beside the step kernel at 2^23 drones (roofline.beyond_mall.copy_ceiling_GBs) */
int fpv_diag_stream_copy_wide(float* dst, const float* src, int64_t n_floats, void* stream);
/* which XCD runs which workgroup: a launch of `blocks` workgroups of the step kernels' size (128 threads) on `stream` of the
 * current device; workgroup b writes the id (0-7, hardware register XCC_ID) of the XCD it was dispatched to into
 * xcd_of_block[b] (device memory, `blocks` words).  The rotation of the traversal keeps a drone block on "its" XCD from launch
 * to launch only if the dispatcher deals workgroups round-robin over the eight XCDs AND starts every launch of a chain on the
 * same XCD - HIP promises neither; this is the probe that watches both (tools/xcd_map_probe.py, bench.py `roofline.xcd_map`). */
int fpv_diag_xcd_map(uint32_t* xcd_of_block, int64_t blocks, void* stream);
/* one wave that does nothing for about `microseconds` (0 < microseconds <= 1000; bounded by the constant-rate clock AND by
 * an iteration count, so every lane leaves) on `stream` of the current device: a kernel of known duration that occupies one
 * CU.  Two streams whose chains of such kernels take as long together as one chain alone run on different hardware
 * queues (fpyv_amd.streams.overlapping_streams); the runtime shares a queue between streams once it has handed out all
 * it has, and chains on a shared queue do not overlap. */
int fpv_diag_busy(double microseconds, void* stream);

/* ---- multi-GPU: contiguous shards, one process (or thread) per GPU, RCCL over xGMI ---------------------------
 * The physics needs no collective (drones are independent); the only exchange of the path is the all-gather of the
 * done mask (and, on request, per-drone episode returns) for a learner that wants the global view.  These entry
 * points give a non-Python host that exchange: RCCL is opened at run time (dlopen: an already loaded librccl - e.g.
 * the one PyTorch ships - is reused, else librccl.so.1 from the loader path or $FPV_RCCL_PATH), so libfpv_hip.so has
 * no link-time dependency on it.  The Python host uses torch.distributed (backend "nccl" = RCCL) instead. */
#define FPV_COMM_ID_BYTES 128
typedef struct fpv_comm* fpv_comm_t;
/* rank 0: create the rendezvous token (ncclGetUniqueId) and hand its 128 bytes to the other ranks out of band */
int fpv_comm_unique_id(uint8_t id[FPV_COMM_ID_BYTES]);
/* every rank: join the communicator (ncclCommInitRank) on `device`; collective - returns when all ranks joined */
int fpv_comm_create(const uint8_t id[FPV_COMM_ID_BYTES], int world_size, int rank, int device, fpv_comm_t* out);
void fpv_comm_destroy(fpv_comm_t c);
/* what the communicator was created with, and the RCCL version (ncclGetVersion: e.g. 22105) actually loaded - lets a
 * benchmark line certify "RCCL saw N ranks"; any out pointer may be NULL */
int fpv_comm_info(fpv_comm_t c, int* world_size, int* rank, int* rccl_version);
/* all-gather of the bit-packed done masks: every rank contributes words_per_rank 64-bit words (its
 * fpv_buffers_t.done_bits, or a whole [steps][words] bucket of them) and receives world_size * words_per_rank words,
 * rank r's block at recv + r * words_per_rank.  Enqueued on `stream`; equal shard sizes on every rank. */
int fpv_allgather_done(fpv_comm_t c, const uint64_t* send_bits, uint64_t* recv_bits, int64_t words_per_rank, void* stream);
/* same for fp32 values (episode returns: fpv_buffers_t.last_return) */
int fpv_allgather_f32(fpv_comm_t c, const float* send, float* recv, int64_t count_per_rank, void* stream);

/* ---- components.PID for N drones at once (src/utils/components.py:15-54) --------------------------------
 * The reference's guidance PID (Drone.force_multiplier_pid, components.py:145,:288): leaky clipped integral,
 * clipped and low-passed derivative, clipped output.  One lane per drone, state as SoA rows
 * pid_state[FPV_PID_ROWS][ld] (fp32, caller-owned).  The same lane function is the racer_pid_variant = 1 rate
 * loop of FPV_MODE_RACER. */
enum { FPV_PID_INTEGRAL = 0, FPV_PID_PREV_DERIVATIVE, FPV_PID_PREV_ERROR, FPV_PID_IS_FIRST, FPV_PID_ROWS };
typedef struct fpv_pid_params {
    uint32_t struct_size;             /* = sizeof(fpv_pid_params_t) */
    uint32_t _reserved;
    double kP, kI, kD, dt;            /* PID.__init__ (components.py:16-20) */
    double integral_clip;             /* default 1   */
    double min_output;                /* default 0.3 */
    double max_output;                /* default 1   */
    double derivative_transition_rate;/* default 0.5 */
} fpv_pid_params_t;
/* PID.reset (components.py:35-41) for the lanes whose mask byte is non-zero (mask NULL = all) */
int fpv_pid_reset(float* pid_state, int64_t ld, int64_t n, const uint8_t* mask, int device, void* stream);
/* PID.__call__(current, target) (components.py:43-54) per drone: out[i] = clip(kP e + kI I + kD D, min, max) with
 * e = current[i] - target, target = target[i] or, when `target` is NULL, target_scalar.  Also writes, when given,
 * error_out[i] (PID.error).  pid_state rows hold integral, prev_derivative, previous_error, is_first. */
int fpv_pid_call(const fpv_pid_params_t* params, float* pid_state, int64_t ld, int64_t n, const float* current,
                 const float* target, float target_scalar, float* out, float* error_out, int device, void* stream);

const char* fpv_last_error(void);
const char* fpv_error_name(int code);
/* Identifier of what stored bits mean, for checkpoints: 0 = the fp16 state storage words (FPV_FLAG_FP16_STATE), 1 = the
 * in-kernel stick-noise stream.  A checkpoint records the string; a library whose string differs cannot continue it bit for
 * bit (fp16 words: cannot decode it at all).  NULL for any other `which`. */
const char* fpv_encoding_id(int which);

#ifdef __cplusplus
}
#endif
#endif /* FPV_ABI_H */
