// fpv_math.h - per-drone (per-lane) fp32 arithmetic of the fused step kernel.
//
// One call advances ONE drone by one step entirely in registers; fpv_hip.hip wraps it with the
// coalesced SoA loads/stores.  The same header is compiled for the host by the test-only lane
// model (oracle/lane_model.cpp) so that fp32 rounding can be studied without a GPU; the product
// never runs it on the CPU.
//
// Arithmetic follows SURVEY.md App. A (paths relative to /root/reference):
//   (1)-(3) stick -> rates / thrust low-pass     src/utils/components.py:185-194
//   (4)     thrust along R[:,2]                  src/utils/kinematics.py:48-49
//   (5)     body-frame quadratic drag, gravity   src/utils/kinematics.py:33-45
//           ground flag on the pre-update pose   src/utils/components.py:235-240
//   (6)     explicit Euler, p first with old v   src/utils/kinematics.py:21-22
//           attitude R <- R*E^T applied TWICE    src/utils/kinematics.py:23,:27-30 + components.py:218
// with the attitude carried as a unit quaternion: R*E^T*E^T  <=>  q (x) conj(q_E)^2.
//
// Rounding discipline: every a*b+c that matters is an explicit fmaf and the translation units are
// built with -ffp-contract=off, so the host lane model and the gfx950 kernel agree bit for bit
// (fpv_sqrt_flushed and '/' are correctly rounded on both).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "fpv_normal_table.h"

#if defined(__HIPCC__)
#define FPV_HD __host__ __device__ __forceinline__
#else
#define FPV_HD static inline
#endif

#define FPV_MATH_FLAG_AUTO_RESET 1u   // = FPV_FLAG_AUTO_RESET (include/fpv_abi.h)
#define FPV_MATH_FLAG_GROUND 2u       // = FPV_FLAG_GROUND
#define FPV_MATH_FLAG_STICK_NOISE 8u  // = FPV_FLAG_STICK_NOISE

struct FpvNoiseK {           // uniform constants of the stick-noise generator
    float tau, omtau;        // transition, 1 - transition   (noise_smooth_test.py:5: 0.1)
    float gain;              // sticks += gain * x_s, then clipped to [-1, 1]
    uint32_t seed_lo, seed_hi;
    uint32_t id_lo, id_hi;   // global id of this shard's drone 0
};

// Constants of one rate loop in arithmetic type T (the Racer as written runs it in float64).
template <class T> struct FpvPidK {
    T dt, inv_dt;
    T dt_over_I[3];
    T gain[3][3];                                            // [axis][kP, kI, kD]
    T integral_clip, min_output, max_output, d_rate, om_d_rate;   // components.PID (pid_variant 1) only
};

// Uniform per-launch constants (kernel argument -> SGPRs).  Derived in double on the host.
struct FpvK {
    float dt;
    float max_rates;        // deg/s
    float kr, omkr;         // rates_transition_rate, 1 - it
    float kt, omkt;         // thrust_transition_rate, 1 - it
    float d3, d2, d1, d0;   // thrust [N] as a cubic in the throttle STICK (x = 50a+50 substituted)
    float dk3, dk2, dk1, dk0;   // the same cubic times thrust_transition_rate: the low-pass is then ONE fma
    float rate_gain, rate_lim;  // -max_rates * rates_transition_rate and its magnitude: clip and gain of the rate low-pass folded
    float inv_mass;
    float g;
    float kdrag_m[3];       // 0.5*rho*Cd_i*A_i / m
    float half_k;           // 0.5 * pi/180 * dt : deg/s -> half-angle per step
    float motor_x[4], motor_y[4];
    float p0[3], v0[3], q0[4];
    float ceiling;
    float goal[3];
    // Racer
    float r_dt, r_inv_mass, r_damp, r_ang_k;
    FpvPidK<float> rf;      // rate-loop constants, fp32
    float motor_radius, ground_k_m, ground_c_m;   // contact distance; spring and damping already divided by m
    float contact_reach;    // largest |motor offset| + motor_radius + 1 mm: centre-distance bound for the object cull
    FpvNoiseK noise;
    uint32_t flags;
    uint32_t r_wide;        // Racer: attitude increment in float64 (Racer.step as written: angle = omega per step)
    uint32_t r_pid_variant; // 0: racer_drone_test.PID.step, 1: components.PID.__call__
    uint32_t angle_mode;    // FPV_ANGLE_*: which sin/cos form the largest possible half-angle of one step needs (fpv_sincos3)
    uint32_t motor_square;  // 1: the motors sit at (+-c, +-c) (the reference's X frame, components.py:120-125): the ground
    float motor_c;          //    flag needs two heights instead of four (see fpv_drone_step_lane)
    double r_ang_k_d;       // r_ang_k in double (the racer_omega_dt variant multiplies by the exact dt)
    FpvPidK<double> rd;     // rate-loop constants in float64 (Racer as written)
};

struct FpvQuat { float w, x, y, z; };

// np.clip(x, lo, hi) for lo <= hi.  On gfx950 ONE v_med3_f32 instead of v_max + v_min; identical for every input,
// NaN included (v_med3 with a NaN operand returns the minimum of the others = lo, and fmaxf(NaN, lo) = lo too).
FPV_HD float fpv_clamp(float x, float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return fminf(fmaxf(x, lo), hi);
#endif
}

// Correctly rounded sqrt of x >= 0 with everything below the smallest normal float flushed to 0 (the arguments are
// sums of squares of speeds / distances: 1e-19 m/s is nothing).  The host build calls sqrtf; the kernel issues
// v_sqrt_f32 (1 ulp) and the compiler's own two-sided correction step, but not the x 2^32 pre-scaling, the un-scaling
// and the zero / infinity class test that `sqrtf` carries for denormal arguments: 11 instructions instead of 16,
// the same bits for every normal x, +0 and +inf.
FPV_HD float fpv_sqrt_flushed(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_sqrtf(x);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = fmaf(-r_dn, r, x), e_up = fmaf(-r_up, r, x);
    float y = e_dn <= 0.0f ? r_dn : r;
    y = e_up > 0.0f ? r_up : y;
    return x < 1.17549435e-38f ? 0.0f : y;
#else
    return x < 1.17549435e-38f ? 0.0f : sqrtf(x);
#endif
}

// sin and cos for |x| <= pi/4 (half-angles of one step's rotation).  Truncation error
// < 3e-9 relative; no range reduction, no slow path.
FPV_HD void fpv_sincos_small(float x, float* s, float* c)
{
    const float x2 = x * x;
    float ps = fmaf(x2, 2.7557319e-6f, -1.9841270e-4f);
    ps = fmaf(ps, x2, 8.3333333e-3f);
    ps = fmaf(ps, x2, -1.6666667e-1f);
    *s = fmaf(x * x2, ps, x);
    float pc = fmaf(x2, -2.7557319e-7f, 2.4801587e-5f);
    pc = fmaf(pc, x2, -1.3888889e-3f);
    pc = fmaf(pc, x2, 4.1666667e-2f);
    pc = fmaf(pc, x2, -0.5f);
    *c = fmaf(x2, pc, 1.0f);
}

// sin and cos of an angle of any size a step can produce (|x| up to ~1e3 rad) in fp32 WITHOUT a library call: two-part
// Cody-Waite reduction by pi/2 (k * C1 is exact inside the fma, so the first step rounds once; C2 = pi/2 - fl(pi/2)
// restores the bits fl() dropped; the neglected third part is k * 1e-15), then the short polynomials on |r| <= pi/4 and
// the quadrant swap.  The same instructions on the host and on gfx950, so the big-angle step, the fp32 Racer and the
// reset kernel agree with the host build bit for bit (round 2 called sincosf here: device libm != host libm).
FPV_HD void fpv_sincos_reduced(float x, float* s, float* c)
{
    const float k = rintf(x * 0.63661977f);                // 2/pi
    float r = fmaf(-k, 1.57079637e+00f, x);                // fl(pi/2) = 0x3fc90fdb
    r = fmaf(-k, -4.37113883e-08f, r);                     // pi/2 - fl(pi/2)
    float sr, cr;
    fpv_sincos_small(r, &sr, &cr);
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cr : sr, cs = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cs : cs;
}

// sin and cos for |x| <= 0.03 (half-angles of one step at dt <= 1/60 s and 200 deg/s): two terms each,
// truncation x^4/120 < 7e-9 relative on sin and x^6/720 < 1e-12 on cos - below fp32 rounding.
FPV_HD void fpv_sincos_tiny(float x, float* s, float* c)
{
    const float x2 = x * x;
    *s = fmaf(x * x2, -1.6666667e-1f, x);
    *c = fmaf(x2, fmaf(x2, 4.1666667e-2f, -0.5f), 1.0f);
}

// The three half-angles of one step's attitude increment.  `mode` is wave-uniform and comes from the host-known bound
// on the half-angle (fpv_derive_constants): 0 = two-term series (<= 0.03 rad), 1 = five-term polynomials (<= pi/4),
// 2 = range reduction first (a step that can turn by more than 90 degrees about an axis).  ONE uniform branch covers
// all three axes; every kernel carries all three forms (round 2 built a second instantiation of every kernel for
// mode 2 and tested `tiny` once per axis).
#define FPV_ANGLE_TINY 0u
#define FPV_ANGLE_SMALL 1u
#define FPV_ANGLE_REDUCED 2u
FPV_HD void fpv_sincos3(uint32_t mode, float x0, float x1, float x2, float s[3], float c[3])
{
    if (mode == FPV_ANGLE_TINY) {
        fpv_sincos_tiny(x0, &s[0], &c[0]); fpv_sincos_tiny(x1, &s[1], &c[1]); fpv_sincos_tiny(x2, &s[2], &c[2]);
    } else if (mode == FPV_ANGLE_SMALL) {
        fpv_sincos_small(x0, &s[0], &c[0]); fpv_sincos_small(x1, &s[1], &c[1]); fpv_sincos_small(x2, &s[2], &c[2]);
    } else {
        fpv_sincos_reduced(x0, &s[0], &c[0]);
        fpv_sincos_reduced(x1, &s[1], &c[1]);
        fpv_sincos_reduced(x2, &s[2], &c[2]);
    }
}

// Rotation matrix columns from a unit quaternion (helper_functions.py:100-117).
struct FpvRot { float r00, r01, r02, r10, r11, r12, r20, r21, r22; };

FPV_HD FpvRot fpv_rot(FpvQuat q)
{
    // 20 instructions (round 3: 27): the doubled components enter every product once, each off-diagonal pair shares one
    // product through an fma, the diagonal shares 1 - 2x^2 and 1 - 2y^2
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2;               // 2x^2, 2y^2, 2z^2
    const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    const float ax = 1.0f - xx, ay = 1.0f - yy;
    FpvRot R;
    R.r00 = ay - zz;                R.r01 = fmaf(q.x, y2, -wz);     R.r02 = fmaf(q.x, z2, wy);
    R.r10 = fmaf(q.x, y2, wz);      R.r11 = ax - zz;                R.r12 = fmaf(q.y, z2, -wx);
    R.r20 = fmaf(q.x, z2, -wy);     R.r21 = fmaf(q.y, z2, wx);      R.r22 = ax - yy;
    return R;
}

// The first two members of Drone.step's return triple (components.py:247-248) for one drone:
//   rt[9]   = rotation_matrix.T, row-major - the attitude AFTER the step;
//   gyro[9] = euler_angles_to_rotation_matrix(*rates) = Rz(rates_z) Ry(rates_y) Rx(rates_x) with the low-passed rates,
//             DEGREES PER SECOND, used as radians (quirk Q6) - tens to hundreds of radians, hence the reduced sin / cos.
// (The third member, R_new @ acc, is the step kernel's `accel` output.)
FPV_HD void fpv_return_matrices(FpvQuat q, float rx, float ry, float rz, float rt[9], float gyro[9])
{
    const FpvRot R = fpv_rot(q);
    rt[0] = R.r00; rt[1] = R.r10; rt[2] = R.r20;
    rt[3] = R.r01; rt[4] = R.r11; rt[5] = R.r21;
    rt[6] = R.r02; rt[7] = R.r12; rt[8] = R.r22;
    float sr, cr, sp, cp, sy, cy;
    fpv_sincos_reduced(rx, &sr, &cr);
    fpv_sincos_reduced(ry, &sp, &cp);
    fpv_sincos_reduced(rz, &sy, &cy);
    gyro[0] = cy * cp; gyro[1] = fmaf(cy * sp, sr, -(sy * cr)); gyro[2] = fmaf(cy * sp, cr, sy * sr);      // helper_functions.py:39-44
    gyro[3] = sy * cp; gyro[4] = fmaf(sy * sp, sr, cy * cr);    gyro[5] = fmaf(sy * sp, cr, -(cy * sr));
    gyro[6] = -sp;     gyro[7] = cp * sr;                       gyro[8] = cp * cr;
}

// q <- normalise(q + q (x) d) for a small-ish quaternion increment d = (dw, dx, dy, dz) (i.e. the
// full factor is 1 + d).  First-order renormalisation: |q|^2 - 1 stays at rounding level because
// it is corrected every step.
// NORM = false leaves the renormalisation out: the fp16-storage kernels renormalise when they widen the stored
// quaternion (every step), and q (x) (1 + d) is unit for unit q anyway - only rounding drift is at stake.
template <bool NORM = true>
FPV_HD FpvQuat fpv_quat_advance(FpvQuat q, float dw, float dx, float dy, float dz)
{
    FpvQuat n;
    n.w = q.w + fmaf(q.w, dw, fmaf(-q.x, dx, fmaf(-q.y, dy, -q.z * dz)));
    n.x = q.x + fmaf(q.w, dx, fmaf(q.x, dw, fmaf(q.y, dz, -q.z * dy)));
    n.y = q.y + fmaf(q.w, dy, fmaf(-q.x, dz, fmaf(q.y, dw, q.z * dx)));
    n.z = q.z + fmaf(q.w, dz, fmaf(q.x, dy, fmaf(-q.y, dx, q.z * dw)));
    if (!NORM) return n;
    const float e = fmaf(n.w, n.w, fmaf(n.x, n.x, fmaf(n.y, n.y, fmaf(n.z, n.z, -1.0f))));
    const float k = fmaf(0.375f * e, e, -0.5f * e);      // 1/sqrt(1+e) - 1
    n.w = fmaf(k, n.w, n.w); n.x = fmaf(k, n.x, n.x); n.y = fmaf(k, n.y, n.y); n.z = fmaf(k, n.z, n.z);
    return n;
}

// Unit quaternion (w,x,y,z) of a rotation matrix m (row-major, body -> world): Shepperd's branch on the largest of
// (trace, m00, m11, m22), then one normalisation.  Used by the guidance override of Drone.step, which REPLACES the
// attitude by a caller-supplied matrix (components.py:230-231); the matrix is expected to be a rotation (the
// reference builds it from normalised cross products, components.py:283-300), a small loss of orthonormality is
// absorbed by the normalisation.
FPV_HD FpvQuat fpv_quat_from_rot(const float m[9])
{
    const float tr = m[0] + m[4] + m[8];
    FpvQuat q;
    if (tr > 0.0f) {
        q.w = 1.0f + tr; q.x = m[7] - m[5]; q.y = m[2] - m[6]; q.z = m[3] - m[1];
    } else if (m[0] >= m[4] && m[0] >= m[8]) {
        q.w = m[7] - m[5]; q.x = 1.0f + m[0] - m[4] - m[8]; q.y = m[1] + m[3]; q.z = m[2] + m[6];
    } else if (m[4] >= m[8]) {
        q.w = m[2] - m[6]; q.x = m[1] + m[3]; q.y = 1.0f + m[4] - m[0] - m[8]; q.z = m[5] + m[7];
    } else {
        q.w = m[3] - m[1]; q.x = m[2] + m[6]; q.y = m[5] + m[7]; q.z = 1.0f + m[8] - m[0] - m[4];
    }
    const float inv = 1.0f / fpv_sqrt_flushed(fmaf(q.w, q.w, fmaf(q.x, q.x, fmaf(q.y, q.y, q.z * q.z))));
    q.w *= inv; q.x *= inv; q.y *= inv; q.z *= inv;
    return q;
}

struct FpvDroneState {
    float px, py, pz, vx, vy, vz;
    FpvQuat q;
    float rx, ry, rz;   // prev_rates, deg/s
    float thrust;       // prev_thrust, N
};

// ------------------------------------------------------------------------------------------------
// fp16 storage (BASELINE config 4: "fp16 state / fp32 integrator"): v, q, prev_rates, prev_thrust
// live in HBM as IEEE binary16, position stays fp32, all arithmetic stays fp32.  The kernel converts
// with the hardware instructions, the host lane model with integer bit manipulation; they agree bit for bit.
//   * low-pass states (rates, thrust) are rounded to nearest-even (v_cvt_f16_f32): their error does not accumulate;
//   * integrator states (v, q) are rounded STOCHASTICALLY: at dt = 1 ms one step's increment is
//     often below half an fp16 ulp (0.03 m/s against ulp 0.016 at 20 m/s; 3e-4 of quaternion against
//     ulp 5e-4), and round-to-nearest would simply stall them.  Unbiased rounding keeps the
//     expected trajectory and turns the stall into a random walk of ~ulp*sqrt(steps).
//     Stochastic rounding = add 13 uniform random bits below the kept mantissa, then round TOWARD ZERO: one integer
//     add per value and ONE v_cvt_pkrtz_f16_f32 per PAIR of values, which also packs the half2 word of the pair row
//     (round 2: a range test, an add, a mask and a select per value, then a round-to-nearest conversion and a
//     separate packing step - 368 VALU instructions per wave against 251 of the fp32 kernel).
// ------------------------------------------------------------------------------------------------
FPV_HD uint32_t fpv_f32_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
FPV_HD float fpv_bits_f32(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// binary16 <-> fp32.  On the device these are the hardware conversions (v_cvt_f32_f16 /
// v_cvt_f16_f32, round-to-nearest-even, subnormals honoured); the host lane model does the same
// arithmetic with integer bit manipulation (checked against numpy.float16 in the tests).
FPV_HD float fpv_f16_to_f32(uint16_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16 f;
    memcpy(&f, &h, 2);
    return (float)f;
#else
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, em = h & 0x7fffu;
    if (em >= 0x7c00u) return fpv_bits_f32(sign | 0x7f800000u | ((em & 0x3ffu) << 13));   // inf / nan
    if (em < 0x0400u) {                                                                    // subnormal: em * 2^-24
        const float m = (float)em * 5.9604644775390625e-08f;
        return fpv_bits_f32(sign | fpv_f32_bits(m));
    }
    return fpv_bits_f32(sign | ((em << 13) + 0x38000000u));
#endif
}

FPV_HD uint16_t fpv_f32_to_f16_rn(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 f = (_Float16)x;
    uint16_t h;
    memcpy(&h, &f, 2);
    return h;
#else
    const uint32_t b = fpv_f32_bits(x), sign = (b >> 16) & 0x8000u;
    uint32_t a = b & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                                // nan
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                               // >= 65520 -> inf
    if (a < 0x38800000u) {                                                                 // below the smallest normal half
        const float t = fpv_bits_f32(a) * 16777216.0f;                                     // value / 2^-24, exact
        uint32_t q = (uint32_t)(t + 0.5f);
        if ((t + 0.5f) == (float)q && (q & 1u)) q -= 1u;                                   // ties to even
        return (uint16_t)(sign | q);
    }
    a += 0xfffu + ((a >> 13) & 1u);
    return (uint16_t)(sign | ((a - 0x38000000u) >> 13));
#endif
}

// fp32 -> binary16 with round TOWARD ZERO (host side of v_cvt_pkrtz_f16_f32): truncation of the mantissa, subnormal
// halves by truncating |x| / 2^-24, finite overflow saturates at the largest finite half (round-toward-zero never
// produces an infinity), inf stays inf.
FPV_HD uint16_t fpv_f32_to_f16_rtz_host(float x)
{
    const uint32_t b = fpv_f32_bits(x), sign = (b >> 16) & 0x8000u;
    const uint32_t a = b & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                                // nan
    if (a == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);                               // inf
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7bffu);                               // >= 65536: largest finite
    if (a < 0x38800000u) return (uint16_t)(sign | (uint32_t)(fpv_bits_f32(a) * 16777216.0f));   // exact scaling, then truncation
    return (uint16_t)(sign | ((a - 0x38000000u) >> 13));
}

// half2 word (low half first) of two fp32 values, both rounded toward zero: ONE instruction on gfx950
FPV_HD uint32_t fpv_pack_pair_rtz(float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __fp16 fpv_h2 __attribute__((ext_vector_type(2)));
    const fpv_h2 p = __builtin_amdgcn_cvt_pkrtz(lo, hi);
    uint32_t w;
    memcpy(&w, &p, 4);
    return w;
#else
    return (uint32_t)fpv_f32_to_f16_rtz_host(lo) | ((uint32_t)fpv_f32_to_f16_rtz_host(hi) << 16);
#endif
}

// the value whose round-toward-zero conversion is the stochastic rounding of x: 13 uniform random bits added below the
// kept mantissa (unbiased for every normal half: E[result] = x; an exactly representable x is never perturbed because
// its low 13 bits are zero and rnd13 < 2^13).  Below 2^-14 - subnormal halves, spacing 6e-8 - the added bits are a
// relative perturbation of at most 2^-10 and the truncation a bias of at most one spacing: irrelevant for the dynamics.
FPV_HD float fpv_sr_arg(float x, uint32_t rnd13) { return fpv_bits_f32(fpv_f32_bits(x) + rnd13); }

FPV_HD uint16_t fpv_f32_to_f16_sr(float x, uint32_t rnd13) { return (uint16_t)fpv_pack_pair_rtz(fpv_sr_arg(x, rnd13), 0.0f); }

// the hash behind the 7 x 13 random bits one step needs
FPV_HD uint32_t fpv_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// stochastic-rounding seed of step index `step` (64-bit launch counter of the handle): base + step for step < 2^32
// (= ABI <= 3), the high word folded in beyond that so that the rounding noise does not repeat either
FPV_HD uint32_t fpv_round_seed(uint32_t base, uint64_t step)
{
    return base + (uint32_t)step + (uint32_t)(step >> 32) * 0x9e3779b1u;
}

// The 11 sixteen-bit words of one drone in storage order - FIVE pair words (low word first) and one single word that
// lives in a row of its own (89 state+io bytes per env-step):
//   w[0] = (vx, vy)  w[1] = (vz, v_low)  w[2] = (qa, qb)  w[3] = (qc, rx)  w[4] = (ry, rz)   t = prev_thrust
// Round 4 spends the same 22 bytes better (round 3: eleven binary16 values, |dq| 6e-3 and |dp|/|p| 8e-3 after 1000 steps -
// both pure random walks of the stochastic rounding of v and q, ulp x sqrt(steps / 6)):
//   * v: binary16 (round toward zero of the stochastically rounded value) PLUS a 5-bit low word per component - the next
//     five mantissa bits, three of them packed into v_low: 15 mantissa bits instead of 10, a 32 times finer grid;
//   * q: SMALLEST THREE.  The component of largest magnitude is dropped (|q| = 1 gives it back, and it is >= 1/2, so the
//     reconstruction is well conditioned), the other three - each <= 1/sqrt(2) in magnitude - are stored as 15-bit
//     fixed point (grid 4.3e-5, eleven times finer than binary16 near 0.7), stochastically rounded; the 2-bit index of the
//     dropped component sits in the spare top bits of qa and qb.  q and -q are the same attitude: the stored sign makes
//     the dropped component positive.
//   * rates and thrust are low-pass states: binary16, round to nearest even, as before.
struct FpvHalfState { uint32_t w[5]; uint16_t t; };

#define FPV_Q3_SCALE 23168.0f            // 15-bit fixed point: 0.70710678 * 23168 = 16382.3 < 2^14
#define FPV_Q3_INV_SCALE 4.31629834e-05f // 1 / 23168

// floor(x) as int32 (v_cvt_flr_i32_f32 on the device)
FPV_HD int32_t fpv_floor_i32(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rd(x);
#else
    return (int32_t)floorf(x);
#endif
}

FPV_HD void fpv_unpack_half(const FpvHalfState& h, FpvDroneState& s)
{
    const uint32_t ext = h.w[1] >> 16;                       // three 5-bit low words of v
    s.vx = fpv_bits_f32(fpv_f32_bits(fpv_f16_to_f32((uint16_t)h.w[0])) | ((ext & 31u) << 8));
    s.vy = fpv_bits_f32(fpv_f32_bits(fpv_f16_to_f32((uint16_t)(h.w[0] >> 16))) | (((ext >> 5) & 31u) << 8));
    s.vz = fpv_bits_f32(fpv_f32_bits(fpv_f16_to_f32((uint16_t)h.w[1])) | (((ext >> 10) & 31u) << 8));
    // 15-bit two's complement fields -> the three stored components
    const float a = (float)((int32_t)(h.w[2] << 17) >> 17) * FPV_Q3_INV_SCALE;
    const float b = (float)((int32_t)(h.w[2] << 1) >> 17) * FPV_Q3_INV_SCALE;
    const float c = (float)((int32_t)(h.w[3] << 17) >> 17) * FPV_Q3_INV_SCALE;
    const uint32_t idx = ((h.w[2] >> 15) & 1u) | ((h.w[2] >> 30) & 2u);
    const float d = fpv_sqrt_flushed(fmaxf(fmaf(-a, a, fmaf(-b, b, fmaf(-c, c, 1.0f))), 0.0f));
    s.q.w = idx == 0u ? d : a;
    s.q.x = idx == 0u ? a : (idx == 1u ? d : b);
    s.q.y = idx <= 1u ? b : (idx == 2u ? d : c);
    s.q.z = idx == 3u ? d : c;
    s.rx = fpv_f16_to_f32((uint16_t)(h.w[3] >> 16));
    s.ry = fpv_f16_to_f32((uint16_t)h.w[4]); s.rz = fpv_f16_to_f32((uint16_t)(h.w[4] >> 16));
    s.thrust = fpv_f16_to_f32(h.t);
}

// one stored quaternion component: floor(a * 2^8 * scale + 8 random bits) >> 8 = stochastic rounding of a * scale to an integer
FPV_HD uint32_t fpv_q3_field(float a, uint32_t rnd8)
{
    // saturated (one v_med3_i32): two tied largest components of a quaternion whose norm drifted above 1 by more than 5e-5
    // would reach 16384, which the 15-bit field reads back as -16384 - a sign flip of that component
    int32_t n = fpv_floor_i32(fmaf(a, FPV_Q3_SCALE * 256.0f, (float)rnd8)) >> 8;
    n = n < -16383 ? -16383 : (n > 16383 ? 16383 : n);
    return (uint32_t)n & 0x7fffu;
}

// what the eleven words mean, as a checkpoint records it (fpv_encoding_id(0); fpyv_amd/env.py STATE_H_ENCODING): change the
// encoding below and this string changes with it
#define FPV_STATE_H_ENCODING_ID "abi5: v f16+5-bit low words, q smallest-three 15-bit fixed point, rates/thrust f16"
FPV_HD void fpv_pack_half(const FpvDroneState& s, uint32_t seed, uint32_t drone, FpvHalfState& h)
{
    // one full-avalanche hash of (seed, drone); the second word by one multiply-xorshift (a bijection of a uniform word:
    // uniform again): 6 x 8 random bits are needed
    const uint32_t r0 = fpv_mix32(seed * 0x9e3779b9u + drone);
    uint32_t r1 = r0 * 0x9e3779b1u; r1 ^= r1 >> 15;
    // v: 8 random bits below the 15 kept mantissa bits, then truncation (round toward zero in both parts)
    const uint32_t bx = fpv_f32_bits(s.vx) + (r0 & 0xffu), by = fpv_f32_bits(s.vy) + ((r0 >> 8) & 0xffu);
    const uint32_t bz = fpv_f32_bits(s.vz) + ((r0 >> 16) & 0xffu);
    const uint32_t ext = ((bx >> 8) & 31u) | (((by >> 8) & 31u) << 5) | (((bz >> 8) & 31u) << 10);
    h.w[0] = fpv_pack_pair_rtz(fpv_bits_f32(bx), fpv_bits_f32(by));
    h.w[1] = fpv_pack_pair_rtz(fpv_bits_f32(bz), 0.0f) | (ext << 16);
    // q: drop the component of largest magnitude, make it positive
    const float aw = fabsf(s.q.w), ax = fabsf(s.q.x), ay = fabsf(s.q.y), az = fabsf(s.q.z);
    const bool b01 = ax > aw, b23 = az > ay;
    const float m01 = b01 ? ax : aw, m23 = b23 ? az : ay;
    const bool top = m23 > m01;
    const uint32_t idx = top ? (b23 ? 3u : 2u) : (b01 ? 1u : 0u);
    const float big = top ? (b23 ? s.q.z : s.q.y) : (b01 ? s.q.x : s.q.w);
    const uint32_t flip = fpv_f32_bits(big) & 0x80000000u;
    const float qa = fpv_bits_f32(fpv_f32_bits(idx == 0u ? s.q.x : s.q.w) ^ flip);
    const float qb = fpv_bits_f32(fpv_f32_bits(idx <= 1u ? s.q.y : s.q.x) ^ flip);
    const float qc = fpv_bits_f32(fpv_f32_bits(idx <= 2u ? s.q.z : s.q.y) ^ flip);
    h.w[2] = fpv_q3_field(qa, r0 >> 24) | ((idx & 1u) << 15) | (fpv_q3_field(qb, r1 & 0xffu) << 16) | ((idx & 2u) << 30);
    h.w[3] = fpv_q3_field(qc, (r1 >> 8) & 0xffu) | ((uint32_t)fpv_f32_to_f16_rn(s.rx) << 16);
    h.w[4] = (uint32_t)fpv_f32_to_f16_rn(s.ry) | ((uint32_t)fpv_f32_to_f16_rn(s.rz) << 16);
    h.t = fpv_f32_to_f16_rn(s.thrust);
}

// ------------------------------------------------------------------------------------------------
// In-kernel stick noise: the profile of tests/noise_smooth_test.py:6-12,
//   x ~ N(0,1);  x_s <- (1 - tau) x_s + tau x,
// generated per drone and channel from a counter-based generator (Philox4x32, Salmon et al., SC'11), so a
// drone's stream depends only on (seed, GLOBAL drone id, step) - not on the batch, lane or shard.
//
// Round 4 rebuilt the generator for cost (418 -> ~275 vector instructions per env-step in the k-step kernel):
//   * Philox4x32-7 instead of -10 (seven rounds pass BigCrush - Random123's "Crush-resistant" minimum; the
//     ten-round function stays for its known-answer test);
//   * four normals from the four words by a table-driven inverse CDF (fpv_normal_from_word) instead of Box-Muller:
//     no logarithm, no division, no square root, no sin/cos - ten instructions and one 16-byte table read per normal.
// The reference's profile is unseeded (np.random.normal, noise_smooth_test.py:8): there is no stream to reproduce, only
// a distribution, so the streams of ABI <= 4 were given up for this (ABI 5).
// ------------------------------------------------------------------------------------------------
template <int ROUNDS>
FPV_HD void fpv_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define FPV_NOISE_PHILOX_ROUNDS 7

// One row of the inverse-CDF table (csrc/fpv_normal_table.h, generated by tools/gen_normal_table.py): 16 bytes, one
// ds_read_b128 on the device (the kernels stage the 2 KB table in LDS), one aligned load on the host.
struct alignas(16) FpvNormalRow { float c0, c1, c2, c3; };
#if !defined(__HIP_DEVICE_COMPILE__)
static const FpvNormalRow fpv_normal_table_host[FPV_NTAB_ROWS] = FPV_NTAB_DATA;
#endif

// One standard normal from one 32-bit word: the top bit is the sign, the other 31 bits (made odd, so never zero) the
// tail probability p = x / 2^32 in (0, 0.5]; |z| = -Phi^-1(p) as a cubic in t = 1 + (the low 21 mantissa bits of x as a
// fraction of the binade) on the table row selected by the exponent and the two leading mantissa bits of x (rows are
// stored rotated so that the index is a shift and a mask; every row's cubic is >= 0, so the sign is OR-ed in: see
// tools/gen_normal_table.py; max |error| 2.2e-6, largest |z| 6.23).  Branch-free, 9 vector instructions + one 16-byte
// read; the same instructions on the host and on gfx950.
FPV_HD float fpv_normal_from_word(uint32_t w, const FpvNormalRow* __restrict__ table)
{
    const uint32_t k = (w & 0x7fffffffu) | 1u;
    const uint32_t b = fpv_f32_bits((float)k);                                    // 2^0 <= x <= 2^31: exponent field 127 .. 158
    const FpvNormalRow c = table[(b >> (23 - FPV_NTAB_SUB_BITS)) & (FPV_NTAB_ROWS - 1)];
    const float t = fpv_bits_f32((b & ((1u << (23 - FPV_NTAB_SUB_BITS)) - 1u)) | 0x3f800000u);          // [1, 1.25), exact
    const float m = fmaf(fmaf(fmaf(c.c3, t, c.c2), t, c.c1), t, c.c0);            // >= 0
    return fpv_bits_f32(fpv_f32_bits(m) | (w & 0x80000000u));
}

// four standard normals of (seed, global drone id, 64-bit step index): one Philox block, one word per normal
FPV_HD void fpv_normal4(uint32_t seed_lo, uint32_t seed_hi, uint32_t drone_lo, uint32_t drone_hi, uint32_t step_lo,
                        uint32_t step_hi, const FpvNormalRow* __restrict__ table, float z[4])
{
    uint32_t r[4];
    fpv_philox4x32<FPV_NOISE_PHILOX_ROUNDS>(drone_lo, drone_hi, step_lo, step_hi, seed_lo, seed_hi, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = fpv_normal_from_word(r[k], table);
}

// advance the EMA state ns[4] and perturb the action in place
// which stream a (seed, drone, step) triple names, as a checkpoint records it (fpv_encoding_id(1); env.py NOISE_GENERATOR)
#define FPV_NOISE_GENERATOR_ID "abi5: philox4x32-7, table-driven inverse normal CDF"
FPV_HD void fpv_stick_noise(const FpvNoiseK& N, uint64_t step, uint64_t local_id, const FpvNormalRow* __restrict__ table,
                            float ns[4], float a[4])
{
    const uint64_t gid = (((uint64_t)N.id_hi << 32) | N.id_lo) + local_id;
    float z[4];
    fpv_normal4(N.seed_lo, N.seed_hi, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step, (uint32_t)(step >> 32), table, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ns[k] = fmaf(z[k], N.tau, ns[k] * N.omtau);
        a[k] = fpv_clamp(fmaf(N.gain, ns[k], a[k]), -1.0f, 1.0f);
    }
}

struct FpvStepOut {
    float ax, ay, az;   // R_new @ acc
    float reward;
    bool done;          // ground flag (reference) OR ceiling (build), before any auto-reset
};

// ------------------------------------------------------------------------------------------------
// object_list collisions (components.py:198-214) for a short ordered table of analytic objects:
// Ground plane z = 0 (:674-680), Cylinder (:710-729, as written) and Target sphere (:774-778).
// Per object: distances of the four motors; if any is negative the drone has crashed and the pass
// stops, keeping the forces of EARLIER objects; otherwise every motor closer than motor_radius adds
// the spring force (-k (d - r_m) - c v.n) n (kinematics.py:56-59).  Forces come back divided by m.
// ------------------------------------------------------------------------------------------------
#ifndef FPV_MAX_OBJECTS
#define FPV_MAX_OBJECTS 8
#endif
struct FpvObject { int32_t type; float x, y, z, radius, height; };      // type: 0 Ground, 1 Cylinder, 2 Target
// `has_ground`, `lo`, `hi`: a conservative summary of the whole list for ONE wave-level test before the per-object
// pass - a drone whose centre is neither within contact reach of the ground plane (if the list has one) nor inside the
// axis-aligned box around every cylinder and sphere (grown by the contact reach) cannot touch anything: filled by
// fpv_objects_bounds() on the host, for the kernel and for the lane model alike.
struct FpvObjects { int32_t count; int32_t has_ground; float lo[3], hi[3]; FpvObject o[FPV_MAX_OBJECTS]; };

// `reach` = K.contact_reach (arm + motor_radius + 1 mm).  The box is grown by a little more than the reach so that
// fp32 rounding of the bounds can only make the test MORE conservative; an empty box (no cylinder / sphere) is lo > hi.
static inline void fpv_objects_bounds(FpvObjects& T, float reach)
{
    T.has_ground = 0;
    for (int c = 0; c < 3; ++c) { T.lo[c] = 3.0e38f; T.hi[c] = -3.0e38f; }
    for (int k = 0; k < T.count; ++k) {
        const FpvObject& ob = T.o[k];
        if (ob.type == 0) { T.has_ground = 1; continue; }
        const float r = fabsf(ob.radius) + reach, m = 1.0e-3f;
        const float cl[3] = {ob.x - r, ob.y - r, ob.type == 1 ? ob.z - reach : ob.z - r};
        const float ch[3] = {ob.x + r, ob.y + r, ob.type == 1 ? ob.z + fabsf(ob.height) + reach : ob.z + r};
        for (int c = 0; c < 3; ++c) {
            const float l = cl[c] - m - 1.0e-6f * fabsf(cl[c]), h = ch[c] + m + 1.0e-6f * fabsf(ch[c]);
            if (l < T.lo[c]) T.lo[c] = l;
            if (h > T.hi[c]) T.hi[c] = h;
        }
    }
}

// wave-level "does any lane need this": on the device one ballot, on the host the lane's own flag
#if defined(__HIP_DEVICE_COMPILE__)
#define FPV_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define FPV_WAVE_ANY(x) (x)
#endif

// "this value is produced here": stops the compiler from hoisting what is computed from it out of a rarely taken
// branch (loop-invariant code motion would otherwise keep those results in registers on the fast path too)
#if defined(__HIP_DEVICE_COMPILE__)
#define FPV_KEEP_HERE(x) asm volatile("" : "+v"(x))
#else
#define FPV_KEEP_HERE(x) ((void)0)
#endif

// R: the pre-update attitude.  The twelve world coordinates of the motors are formed per object INSIDE the branch
// that a wave takes only when one of its drones is near that object (same expressions, same bits as forming them
// up front): on the culled fast path the kernel keeps the register budget of the plain step kernel (round 2 formed
// them before the object loop: 102 VGPRs, 4 waves per SIMD, even when every object was culled).
FPV_HD bool fpv_collide_objects(const FpvK& K, const FpvObjects& T, const FpvRot& R, float cx, float cy, float cz,
                                float vx, float vy, float vz, float acc[3])
{
    bool crashed = false;
    acc[0] = acc[1] = acc[2] = 0.0f;
    const float reach = K.contact_reach;       // arm + motor_radius + 1 mm: no motor can touch beyond this
    // one test for the whole list first (see FpvObjects): the common wave - nobody near anything - leaves here with a
    // dozen instructions and ONE ballot instead of a cull and a ballot per object (1.7 us of a 24 us kernel in round 2)
    {
        const bool maybe = (T.has_ground != 0 && cz < reach)
                           || (cx > T.lo[0] && cx < T.hi[0] && cy > T.lo[1] && cy < T.hi[1] && cz > T.lo[2] && cz < T.hi[2]);
        if (!FPV_WAVE_ANY(maybe)) return false;
    }
    for (int o = 0; o < T.count && !crashed; ++o) {
        const FpvObject& ob = T.o[o];
        // cheap conservative cull on the drone centre; a whole wave with nobody near skips the
        // sqrt/divide-heavy per-motor pass (culled lanes would get exactly zero force and no crash)
        bool near;
        {
            const float dx = cx - ob.x, dy = cy - ob.y, dz = cz - ob.z;
            const float rr = ob.radius + reach;
            if (ob.type == 0) near = cz < reach;
            else if (ob.type == 1) near = fmaf(dx, dx, dy * dy) < rr * rr && dz > -reach && dz < ob.height + reach;
            else near = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < rr * rr;
        }
        if (!FPV_WAVE_ANY(near)) continue;
        float mx[4], my[4], mz[4];
        {
            float px = cx, py = cy, pz = cz;
            FPV_KEEP_HERE(px); FPV_KEEP_HERE(py); FPV_KEEP_HERE(pz);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                mx[m] = px + fmaf(K.motor_x[m], R.r00, K.motor_y[m] * R.r01);
                my[m] = py + fmaf(K.motor_x[m], R.r10, K.motor_y[m] * R.r11);
                mz[m] = pz + fmaf(K.motor_x[m], R.r20, K.motor_y[m] * R.r21);
            }
        }
        // distances of all four motors first (the crash test needs them all, components.py:203-206); the radial
        // length rr is kept, the NORMAL (a correctly rounded division per motor) is formed only for a motor that is
        // actually in contact - the same arithmetic in the same order, evaluated lazily
        float dist[4], rrm[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float rx = mx[m] - ob.x, ry = my[m] - ob.y, rz = mz[m] - ob.z;
            if (ob.type == 0) {
                dist[m] = mz[m]; rrm[m] = 1.0f;
            } else if (ob.type == 1) {
                const float rr = fpv_sqrt_flushed(fmaf(rx, rx, ry * ry));
                const float d2 = rr - ob.radius;
                const float top = ob.z + ob.height;
                if (ob.z < mz[m] && mz[m] < top) dist[m] = d2;
                else {
                    const float dh = fminf(fabsf(mz[m] - ob.z), fabsf(mz[m] - top));
                    dist[m] = fpv_sqrt_flushed(fmaf(d2, d2, dh * dh));
                }
                rrm[m] = rr;
            } else {
                const float rr = fpv_sqrt_flushed(fmaf(rx, rx, fmaf(ry, ry, rz * rz)));
                dist[m] = rr - ob.radius;
                rrm[m] = rr;
            }
        }
        if (dist[0] < 0.0f || dist[1] < 0.0f || dist[2] < 0.0f || dist[3] < 0.0f) { crashed = true; break; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float d = dist[m] - K.motor_radius;
            if (d < 0.0f) {
                const float rx = mx[m] - ob.x, ry = my[m] - ob.y, rz = mz[m] - ob.z;
                float nx = 0.0f, ny = 0.0f, nz = 1.0f;             // Ground: +z
                if (ob.type == 1) {
                    const float top = ob.z + ob.height;
                    if (ob.z < rz && rz < top) {                   // relative z against absolute bounds, as written
                        const float inv = 1.0f / rrm[m];
                        nx = rx * inv; ny = ry * inv; nz = 0.0f;
                    } else {
                        nz = (fabsf(rz - ob.z) < fabsf(rz - top)) ? -1.0f : 1.0f;
                    }
                } else if (ob.type != 0) {
                    const float inv = 1.0f / rrm[m];
                    nx = rx * inv; ny = ry * inv; nz = rz * inv;
                }
                const float vn = fmaf(vx, nx, fmaf(vy, ny, vz * nz));
                const float f = fmaf(-K.ground_k_m, d, -K.ground_c_m * vn);
                acc[0] = fmaf(f, nx, acc[0]); acc[1] = fmaf(f, ny, acc[1]); acc[2] = fmaf(f, nz, acc[2]);
            }
        }
    }
    return crashed;
}

// OUT = false skips the values that only leave the lane (R_new @ acc and the reward): the k-step kernel
// needs them only on the steps whose outputs are stored.  The state update is identical either way.
// SQ = true is a promise of the caller: K.motor_square is set and FPV_MATH_FLAG_GROUND is not (the k-step kernels
// test that once per launch); the ground flag then comes from two motor heights instead of four, bit for bit the
// same flag.  The single-step kernels are HBM-bound and keep the four-height form (SQ = false): there the extra
// uniform test costs more than the 14 instructions it saves.
// NORM = false: no renormalisation of the advanced quaternion (fp16-storage kernels only, see fpv_quat_advance).
template <bool OBJ = false, bool OUT = true, bool SQ = false, bool NORM = true>
FPV_HD FpvStepOut fpv_drone_step_lane(const FpvK& K, FpvDroneState& s, float a0, float a1, float a2, float a3,
                                      float wx, float wy, float wz, const FpvObjects* objs = nullptr,
                                      float* kahan = nullptr, const float* rot_over = nullptr, float thrust_over = 0.0f)
{
    // (1)-(2) stick -> rate command (deg/s), clipped, low-passed          components.py:185-189
    // clip(-a max, +-max) kr = clip(a (-max kr), +-(max kr)): the filter gain folded into the clip's operands (host, in
    // double), so that an axis is mul + med3 + fma instead of mul + med3 + mul + fma
    const float c0 = fpv_clamp(a0 * K.rate_gain, -K.rate_lim, K.rate_lim);
    const float c1 = fpv_clamp(a1 * K.rate_gain, -K.rate_lim, K.rate_lim);
    const float c2 = fpv_clamp(a2 * K.rate_gain, -K.rate_lim, K.rate_lim);
    s.rx = fmaf(s.rx, K.omkr, c0);
    s.ry = fmaf(s.ry, K.omkr, c1);
    s.rz = fmaf(s.rz, K.omkr, c2);
    // (3) thrust cubic (Horner in the stick, no clamp), low-passed; the cubic carries the filter gain   components.py:136,:192-194
    s.thrust = fmaf(s.thrust, K.omkt, fmaf(fmaf(fmaf(K.dk3, a3, K.dk2), a3, K.dk1), a3, K.dk0));
    // guidance override (components.py:230-232): AFTER action2force has advanced prev_rates / prev_thrust the
    // attitude is replaced by the caller's rotation_matrix and the thrust becomes thrust_force * R[:,2]; drag, the
    // motor positions and the attitude increment then start from the new attitude.  A NaN thrust_force leaves
    // this drone alone (the reference's `rotation_matrix is None` for one drone of a batch).
    float thrust_now = s.thrust;
    if (rot_over && thrust_over == thrust_over) {
        s.q = fpv_quat_from_rot(rot_over);
        thrust_now = thrust_over;
    }

    const FpvRot R = fpv_rot(s.q);                                         // PRE-update attitude

    // (5) drag: F_b = -k (R^T v_s) |v_s|, back to world; wind is ADDED      kinematics.py:33-38
    const float ux = s.vx + wx, uy = s.vy + wy, uz = s.vz + wz;
    const float speed = fpv_sqrt_flushed(fmaf(ux, ux, fmaf(uy, uy, uz * uz)));
    const float bx = fmaf(R.r00, ux, fmaf(R.r10, uy, R.r20 * uz));
    const float by = fmaf(R.r01, ux, fmaf(R.r11, uy, R.r21 * uz));
    const float bz = fmaf(R.r02, ux, fmaf(R.r12, uy, R.r22 * uz));
    const float fx = -K.kdrag_m[0] * bx * speed;
    const float fy = -K.kdrag_m[1] * by * speed;
    // (4) thrust along body z (third column), gravity; everything already divided by m.  Thrust and the z component of
    // the drag are both along the body z axis: they are added in the body frame and rotated together
    const float fz = fmaf(-K.kdrag_m[2] * bz, speed, thrust_now * K.inv_mass);
    float accx = fmaf(R.r00, fx, fmaf(R.r01, fy, R.r02 * fz));
    float accy = fmaf(R.r10, fx, fmaf(R.r11, fy, R.r12 * fz));
    float accz = fmaf(R.r20, fx, fmaf(R.r21, fy, fmaf(R.r22, fz, -K.g)));

    // ground flag: any motor below z = 0 on the PRE-update pose             components.py:235-240
    // ground contact (object_list = [Ground]): each motor closer than motor_radius to z = 0 adds a
    // spring force along +z; if ANY motor is below the plane the reference reports a crash and
    // returns before adding any force                                        components.py:198-214
    bool done = false;
    if (SQ && !OBJ) {
        // the reference's X frame, motors at (+-c, +-c): the four heights above the centre
        // are +-hA and +-hB with hA = fl(c r20 + fl(c r21)), hB = fl(-c r20 + fl(c r21)) (negating both products
        // negates the rounded result exactly), so "any of fl(pz +- hA), fl(pz +- hB) below zero" is exactly
        // "pz < max(|hA|, |hB|)": 5 instructions instead of 19, the same flag bit for bit.
        const float t = K.motor_c * R.r21;
        const float hA = fmaf(K.motor_c, R.r20, t), hB = fmaf(-K.motor_c, R.r20, t);
        done = s.pz < fmaxf(fabsf(hA), fabsf(hB));
    } else {
        float mzg[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            mzg[m] = s.pz + fmaf(K.motor_x[m], R.r20, K.motor_y[m] * R.r21);
            done = done || (mzg[m] < 0.0f);
        }
        if (K.flags & FPV_MATH_FLAG_GROUND) {                // wave-uniform: the spring pass is skipped without the flag
            float contact = 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float d = mzg[m] - K.motor_radius;
                if (d < 0.0f) contact += fmaf(-K.ground_k_m, d, -K.ground_c_m * s.vz);   // kinematics.py:56-59, normal = +z
            }
            if (!done) accz += contact;
        }
    }
    if (OBJ) {                                               // general object_list replaces the ground-only pass
        float ca[3];
        const bool crashed = fpv_collide_objects(K, *objs, R, s.px, s.py, s.pz, s.vx, s.vy, s.vz, ca);
        accx += ca[0]; accy += ca[1]; accz += ca[2];
        done = done || crashed;
    }

    // (6) explicit Euler: p with the OLD v, then v                          kinematics.py:21-22
    if (kahan) {
        // compensated (Kahan) accumulation of p += v dt: after ~10^4 steps a plain fp32 sum of 0.02 m
        // increments into a 200 m coordinate has lost 1e-4 of it; the running compensation keeps the
        // fp32 position within ~1 ulp of the exactly accumulated sum (BASELINE config 1, 10 000 steps).
        const float inc[3] = {s.vx, s.vy, s.vz};
        float* pp[3] = {&s.px, &s.py, &s.pz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float y = fmaf(inc[k], K.dt, -kahan[k]);
            const float t = *pp[k] + y;
            kahan[k] = (t - *pp[k]) - y;
            *pp[k] = t;
        }
    } else {
        s.px = fmaf(s.vx, K.dt, s.px); s.py = fmaf(s.vy, K.dt, s.py); s.pz = fmaf(s.vz, K.dt, s.pz);
    }
    if (kahan) {                                             // same compensation for v += acc dt (kahan[3..5])
        const float inc[3] = {accx, accy, accz};
        float* vv[3] = {&s.vx, &s.vy, &s.vz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float y = fmaf(inc[k], K.dt, -kahan[3 + k]);
            const float t = *vv[k] + y;
            kahan[3 + k] = (t - *vv[k]) - y;
            *vv[k] = t;
        }
    } else {
        s.vx = fmaf(accx, K.dt, s.vx); s.vy = fmaf(accy, K.dt, s.vy); s.vz = fmaf(accz, K.dt, s.vz);
    }

    // attitude: q <- q (x) conj(q_E)^2, q_E = qz(psi) qy(theta) qx(phi)     kinematics.py:27-30 (x2)
    float sn[3], cs[3];
    fpv_sincos3(K.angle_mode, s.rx * K.half_k, s.ry * K.half_k, s.rz * K.half_k, sn, cs);
    const float sr = sn[0], cr = cs[0], sp = sn[1], cp = cs[1], sy = sn[2], cy = cs[2];
    const float ew = fmaf(cy * cp, cr, sy * sp * sr);
    const float ex = fmaf(cy * cp, sr, -(sy * sp * cr));
    const float ey = fmaf(cy * sp, cr, sy * cp * sr);
    const float ez = fmaf(sy * cp, cr, -(cy * sp * sr));
    // conj(q_E)^2 - 1 = (-2|e_v|^2, -2 e_w e_v)
    const float vv = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
    const float m2w = -2.0f * ew;
    s.q = fpv_quat_advance<NORM>(s.q, -2.0f * vv, m2w * ex, m2w * ey, m2w * ez);

    FpvStepOut o;
    o.ax = o.ay = o.az = 0.0f; o.reward = 0.0f;
    if (OUT) {
        const FpvRot Rn = fpv_rot(s.q);                                    // components.py:248
        o.ax = fmaf(Rn.r00, accx, fmaf(Rn.r01, accy, Rn.r02 * accz));
        o.ay = fmaf(Rn.r10, accx, fmaf(Rn.r11, accy, Rn.r12 * accz));
        o.az = fmaf(Rn.r20, accx, fmaf(Rn.r21, accy, Rn.r22 * accz));
        const float gx = s.px - K.goal[0], gy = s.py - K.goal[1], gz = s.pz - K.goal[2];
        o.reward = -fpv_sqrt_flushed(fmaf(gx, gx, fmaf(gy, gy, gz * gz)));
    }
    o.done = done || !(fabsf(s.pz) <= K.ceiling);
    return o;
}

// Drone.reset's attitude for a per-drone `ypr` argument (components.py:150-154): the triple is consumed as (roll,
// pitch, yaw) in DEGREES and R = Rz(yaw) Ry(pitch) Rx(roll), i.e. q = qz (x) qy (x) qx of the half angles.  The
// reset kernel and the host build run these instructions (range-reduced sin/cos, no library call).
FPV_HD FpvQuat fpv_quat_from_rpy_deg(float roll_deg, float pitch_deg, float yaw_deg)
{
    const float d2r_half = 0.5f * 0.017453292519943295f;
    float sr, cr, sp, cp, sy, cy;
    fpv_sincos_reduced(roll_deg * d2r_half, &sr, &cr);
    fpv_sincos_reduced(pitch_deg * d2r_half, &sp, &cp);
    fpv_sincos_reduced(yaw_deg * d2r_half, &sy, &cy);
    FpvQuat q;
    q.w = fmaf(cy * cp, cr, sy * sp * sr);
    q.x = fmaf(cy * cp, sr, -(sy * sp * cr));
    q.y = fmaf(cy * sp, cr, sy * cp * sr);
    q.z = fmaf(sy * cp, cr, -(cy * sp * sr));
    return q;
}

FPV_HD void fpv_drone_reset_lane(const FpvK& K, FpvDroneState& s)
{
    s.px = K.p0[0]; s.py = K.p0[1]; s.pz = K.p0[2];
    s.vx = K.v0[0]; s.vy = K.v0[1]; s.vz = K.v0[2];
    s.q.w = K.q0[0]; s.q.x = K.q0[1]; s.q.y = K.q0[2]; s.q.z = K.q0[3];
    s.rx = s.ry = s.rz = 0.0f;
    s.thrust = 0.0f;
}

// sin and cos of an UNBOUNDED angle in float64, the same instruction sequence on the host and on
// gfx950 (explicit fma, no library call): three-part Cody-Waite reduction by pi/2 (each part has
// 33 significant bits, so k * part is exact for |k| < 2^20, i.e. |x| < 1.6e6 rad) and the Taylor
// polynomials to x^15 / x^16 (truncation < 5e-17 on |r| <= pi/4).  Used by the Racer as written,
// whose per-step angle is omega in RADIANS PER STEP (racer_drone_test.py:99): tens of radians, the
// same value step after step, so any fp32 rounding of the increment repeats coherently.
FPV_HD void fpv_sincos_wide(double x, double* s, double* c)
{
    const double k = rint(x * 0.6366197723675814);                 // 2/pi
    double r = fma(-k, 1.57079632673412561417e+00, x);              // pi/2, bits 1..33
    r = fma(-k, 6.07710050630396597660e-11, r);                     //       bits 34..66
    r = fma(-k, 2.02226624871116645580e-21, r);                     //       bits 67..99
    const double r2 = r * r;
    double ps = fma(r2, -7.647163731819816e-13, 1.6059043836821613e-10);     // -1/15!, 1/13!
    ps = fma(ps, r2, -2.505210838544172e-08);                                   // -1/11!
    ps = fma(ps, r2, 2.7557319223985893e-06);                                    //  1/9!
    ps = fma(ps, r2, -1.984126984126984e-04);                                   // -1/7!
    ps = fma(ps, r2, 8.333333333333333e-03);                                    //  1/5!
    ps = fma(ps, r2, -1.6666666666666666e-01);                                   // -1/3!
    const double sr = fma(r * r2, ps, r);
    double pc = fma(r2, 4.779477332387385e-14, -1.1470745597729725e-11);     //  1/16!, -1/14!
    pc = fma(pc, r2, 2.08767569878681e-09);                                    //  1/12!
    pc = fma(pc, r2, -2.755731922398589e-07);                                   // -1/10!
    pc = fma(pc, r2, 2.48015873015873e-05);                                    //  1/8!
    pc = fma(pc, r2, -1.388888888888889e-03);                                   // -1/6!
    pc = fma(pc, r2, 4.1666666666666664e-02);                                    //  1/4!
    pc = fma(pc, r2, -0.5);
    const double cr = fma(r2, pc, 1.0);
    const int q = (int)((long long)k & 3);
    const double ss = (q & 1) ? cr : sr, cs = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cs : cs;
}

// ------------------------------------------------------------------------------------------------
// Racer: rate PID -> torque -> omega -> attitude; thrust along body z, damped velocity.
//   PID.step            tests/racer_drone_test.py:22-32          (pid_variant 0)
//   PID.__call__        src/utils/components.py:43-54            (pid_variant 1: leaky clipped integral,
//                                                                 clipped + low-passed derivative, clipped output)
//   Racer.step          tests/racer_drone_test.py:95-103
// ------------------------------------------------------------------------------------------------
struct FpvRacerState {
    float px, py, pz, vx, vy, vz;
    FpvQuat q;
    float w[3], ierr[3], lerr[3];
    float first;
    float wlo[3], ilo[3];   // low words of omega and of the PID integral (Racer as written only; rows FPV_R_OMEGA_LO.., FPV_R_IERR_LO..)
    float dflt[3];          // components.PID prev_derivative (pid_variant 1 only; rows FPV_R_DFILT..)
};

FPV_HD float fpv_fma_t(float a, float b, float c) { return fmaf(a, b, c); }
FPV_HD double fpv_fma_t(double a, double b, double c) { return fma(a, b, c); }
FPV_HD float fpv_clip_t(float x, float lo, float hi) { return fpv_clamp(x, lo, hi); }                // np.clip
FPV_HD double fpv_clip_t(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// One axis of the rate loop in arithmetic type T (float, or double for the Racer as written).
// Returns the torque; advances integral / last error / filtered derivative.
template <class T, int PIDV>
FPV_HD T fpv_pid_axis(const FpvPidK<T>& P, int i, T actual, T desired, bool first, T& integ, T& last, T& dflt)
{
    if (PIDV == 0) {                                                        // racer_drone_test.py:22-32
        const T err = desired - actual;                                     // :23
        integ = fpv_fma_t(err, P.dt, integ);                                // :25
        const T derr = first ? (T)0 : (err - last) * P.inv_dt;              // :26-29
        last = err;                                                         // :31
        return fpv_fma_t(P.gain[i][0], err, fpv_fma_t(P.gain[i][1], integ, P.gain[i][2] * derr));   // :32
    } else {                                                                // components.py:43-54
        const T err = actual - desired;                                     // :44  current - target
        integ = fpv_clip_t(fpv_fma_t((T)0.99, integ, err * P.dt), -P.integral_clip, P.integral_clip);   // :46
        T d = fpv_clip_t(first ? (T)0 : (err - last) * P.inv_dt, (T)-1, (T)1);                           // :48
        d = fpv_fma_t(P.om_d_rate, dflt, P.d_rate * d);                     // :49
        dflt = d;                                                           // :50
        last = err;                                                         // :53
        return fpv_clip_t(fpv_fma_t(P.gain[i][0], err, fpv_fma_t(P.gain[i][1], integ, P.gain[i][2] * d)),
                          P.min_output, P.max_output);                      // :54
    }
}

// WIDE: the rate loop, omega and the attitude increment in float64 (state rows stay fp32; omega and
// the PID integral are fp32 (hi, lo) pairs).  Needed for Racer.step AS WRITTEN, which turns the
// attitude by omega RADIANS per step (racer_drone_test.py:99): an error d in omega is d radians of
// attitude EVERY step, an fp32 ulp of omega ~ 1 is already 1.2e-7 rad, fp32 gains are off by 6e-8
// relative, and with omega constant the increment quaternion is the same every step so its fp32
// rounding would add up coherently (6e-5 over 1000 steps).  In float64 only the final rounding of q
// to fp32 remains, which is incoherent: measured 7e-7 on the reference captures G7/G8 over the whole
// 1000-step trajectory.  The racer_omega_dt variant (angle = omega*dt) is well conditioned in fp32.
// A fresh, opaque view of a uniform object that lives in the kernel-argument segment (device only; `r` MUST be a
// reference into that segment - the k-step kernels' FpvRollArgs - never a by-value kernel parameter, whose address the
// compiler could only take by copying it to scratch).  What is read through the result is loaded here, not carried
// from an earlier load of the same field: it bounds how long a group of uniforms stays in SGPRs.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ const T& fpv_uniform_again(const T& r)
{
    typedef const __attribute__((address_space(4))) T* P4;
    P4 p = (P4)(&r);
    asm volatile("" : "+s"(p));
    return *(const T*)p;
}
#else
template <class T> static inline const T& fpv_uniform_again(const T& r) { return r; }
#endif

// VIEWS (k-step kernels only, see fpv_uniform_again): the float64 rate loop reads its constants per AXIS - the as-written
// Racer carries 45 double-precision uniforms (90 SGPRs), which together with the output pointers of a step that stores
// reward / done did not fit the SGPR file (10-18 spilled in round 3).
template <bool WIDE, int PIDV = 0, bool OUT = true, bool VIEWS = false>
FPV_HD float fpv_racer_step_lane(const FpvK& K, FpvRacerState& s, float a0, float a1, float a2, float a3)
{
    const float act[3] = {a0, a1, a2};
    const bool first = s.first != 0.0f;
    if (WIDE) {
        double ang[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const FpvPidK<double>& Rd = VIEWS ? fpv_uniform_again(K.rd) : K.rd;
            const double w = (double)s.w[i] + (double)s.wlo[i];
            double integ = (double)s.ierr[i] + (double)s.ilo[i], last = (double)s.lerr[i], df = (double)s.dflt[i];
            const double tq = fpv_pid_axis<double, PIDV>(Rd, i, w, (double)act[i], first, integ, last, df);
            const double wn = fma(tq, Rd.dt_over_I[i], w);                  // :98
            s.ierr[i] = (float)integ; s.ilo[i] = (float)(integ - (double)s.ierr[i]);
            s.lerr[i] = (float)last; s.dflt[i] = (float)df;
            s.w[i] = (float)wn; s.wlo[i] = (float)(wn - (double)s.w[i]);
            ang[i] = 0.5 * wn * K.r_ang_k_d;
        }
        // :99  q <- q (x) qx(a) (x) qy(b) (x) qz(c)   (intrinsic "XYZ"), angle = omega * r_ang_k
        double sa, ca, sb, cb, sc, cc;
        fpv_sincos_wide(ang[0], &sa, &ca);
        fpv_sincos_wide(ang[1], &sb, &cb);
        fpv_sincos_wide(ang[2], &sc, &cc);
        const double w1 = ca * cb, x1 = sa * cb, y1 = ca * sb, z1 = sa * sb;
        const double dw = fma(w1, cc, -z1 * sc), dx = fma(x1, cc, y1 * sc);
        const double dy = fma(y1, cc, -x1 * sc), dz = fma(z1, cc, w1 * sc);
        const double qw = s.q.w, qx = s.q.x, qy = s.q.y, qz = s.q.z;
        const double nw = fma(qw, dw, fma(-qx, dx, fma(-qy, dy, -qz * dz)));
        const double nx = fma(qw, dx, fma(qx, dw, fma(qy, dz, -qz * dy)));
        const double ny = fma(qw, dy, fma(-qx, dz, fma(qy, dw, qz * dx)));
        const double nz = fma(qw, dz, fma(qx, dy, fma(-qy, dx, qz * dw)));
        // |n|^2 = 1 + e with |e| ~ 1e-7 (the fp32 storage of q): 1/sqrt(1+e) = 1 - e/2 + 3e^2/8
        const double e = fma(nw, nw, fma(nx, nx, fma(ny, ny, fma(nz, nz, -1.0))));
        const double k = fma(0.375 * e, e, -0.5 * e);
        s.q.w = (float)fma(k, nw, nw); s.q.x = (float)fma(k, nx, nx);
        s.q.y = (float)fma(k, ny, ny); s.q.z = (float)fma(k, nz, nz);
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tq = fpv_pid_axis<float, PIDV>(K.rf, i, s.w[i], act[i], first, s.ierr[i], s.lerr[i], s.dflt[i]);
            s.w[i] = fmaf(tq, K.rf.dt_over_I[i], s.w[i]);                   // :98
        }
        float sa, ca, sb, cb, sc, cc;
        fpv_sincos_reduced(0.5f * s.w[0] * K.r_ang_k, &sa, &ca);
        fpv_sincos_reduced(0.5f * s.w[1] * K.r_ang_k, &sb, &cb);
        fpv_sincos_reduced(0.5f * s.w[2] * K.r_ang_k, &sc, &cc);
        // qx*qy = (ca cb, sa cb, ca sb, sa sb); then * qz
        const float w1 = ca * cb, x1 = sa * cb, y1 = ca * sb, z1 = sa * sb;
        const float dw = fmaf(w1, cc, -z1 * sc), dx = fmaf(x1, cc, y1 * sc);
        const float dy = fmaf(y1, cc, -x1 * sc), dz = fmaf(z1, cc, w1 * sc);
        FpvQuat n;
        n.w = fmaf(s.q.w, dw, fmaf(-s.q.x, dx, fmaf(-s.q.y, dy, -s.q.z * dz)));
        n.x = fmaf(s.q.w, dx, fmaf(s.q.x, dw, fmaf(s.q.y, dz, -s.q.z * dy)));
        n.y = fmaf(s.q.w, dy, fmaf(-s.q.x, dz, fmaf(s.q.y, dw, s.q.z * dx)));
        n.z = fmaf(s.q.w, dz, fmaf(s.q.x, dy, fmaf(-s.q.y, dx, s.q.z * dw)));
        const float inv = 1.0f / fpv_sqrt_flushed(fmaf(n.w, n.w, fmaf(n.x, n.x, fmaf(n.y, n.y, n.z * n.z))));
        s.q.w = n.w * inv; s.q.x = n.x * inv; s.q.y = n.y * inv; s.q.z = n.z * inv;
    }
    s.first = 0.0f;
    // :100-103 force along the NEW body z, damped velocity, v-then-p
    const FpvRot R = fpv_rot(s.q);
    const float f = a3 * K.r_inv_mass * K.r_dt;
    s.vx = fmaf(K.r_damp, s.vx, f * R.r02);
    s.vy = fmaf(K.r_damp, s.vy, f * R.r12);
    s.vz = fmaf(K.r_damp, s.vz, f * R.r22);
    s.px = fmaf(s.vx, K.r_dt, s.px); s.py = fmaf(s.vy, K.r_dt, s.py); s.pz = fmaf(s.vz, K.r_dt, s.pz);
    if (!OUT) return 0.0f;                           // a k-step launch's quiet steps store no reward
    const float gx = s.px - K.goal[0], gy = s.py - K.goal[1], gz = s.pz - K.goal[2];
    return -fpv_sqrt_flushed(fmaf(gx, gx, fmaf(gy, gy, gz * gz)));
}

FPV_HD void fpv_racer_reset_lane(FpvRacerState& s)
{
    s.px = s.py = s.pz = s.vx = s.vy = s.vz = 0.0f;
    s.q.w = 1.0f; s.q.x = s.q.y = s.q.z = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.w[i] = 0.0f; s.ierr[i] = 0.0f; s.lerr[i] = 0.0f; s.wlo[i] = 0.0f; s.ilo[i] = 0.0f; s.dflt[i] = 0.0f; }
    s.first = 1.0f;
}
