// fpv_derive.h - host-side narrowing of fpv_params_t (double) to the kernel constants FpvK (fp32).
//
// Everything that can be folded on the host in double precision is folded here, once per
// fpv_create/fpv_set_params, so the per-lane fp32 arithmetic starts from correctly rounded
// constants:
//   * the thrust cubic over throttle PERCENT (components.py:136, x = 100(a+1)/2) is re-expanded
//     around the stick value a (x = 50a + 50): same polynomial, but |a| <= 1 keeps the fp32 Horner
//     terms the size of the result instead of cancelling ~90 N against ~-36 N;
//   * 0.5*rho*Cd*A/m per body axis (kinematics.py:36 followed by components.py:243);
//   * deg/s -> half-angle per step, 0.5 * pi/180 * dt (kinematics.py:29).
#pragma once

#include <math.h>
#include <string.h>

#include "../../include/fpv_abi.h"
#include "fpv_math.h"

// Returns 0 or an FPV_E* code; *why receives a static message on failure.
static inline int fpv_derive_constants(const fpv_params_t* P, FpvK* K, const char** why)
{
    *why = "";
    if (P->struct_size != sizeof(fpv_params_t)) { *why = "fpv_params_t.struct_size does not match this library"; return FPV_EINVAL; }
    if (P->mode != FPV_MODE_DRONE && P->mode != FPV_MODE_RACER) { *why = "unknown mode"; return FPV_EINVAL; }
    if ((P->flags & FPV_FLAG_FP16_STATE) && P->mode != FPV_MODE_DRONE) { *why = "FPV_FLAG_FP16_STATE is a drone-mode layout"; return FPV_EINVAL; }
    if ((P->flags & FPV_FLAG_STICK_NOISE) && (P->mode != FPV_MODE_DRONE || (P->flags & FPV_FLAG_FP16_STATE))) { *why = "FPV_FLAG_STICK_NOISE needs drone mode with fp32 state"; return FPV_EINVAL; }
    if ((P->flags & FPV_FLAG_STICK_NOISE) && !(P->noise_transition > 0 && P->noise_transition <= 1)) { *why = "noise_transition must be in (0, 1]"; return FPV_EPARAM; }
    if (!(P->dt > 0) || !isfinite(P->dt)) { *why = "dt must be positive and finite"; return FPV_EPARAM; }
    if (!(P->mass > 0)) { *why = "mass must be positive"; return FPV_EPARAM; }
    if (!(P->max_rates >= 0) || !isfinite(P->max_rates)) { *why = "max_rates must be finite and >= 0"; return FPV_EPARAM; }
    if (P->mode == FPV_MODE_RACER) {
        if (!(P->racer_mass > 0)) { *why = "racer_mass must be positive"; return FPV_EPARAM; }
        for (int i = 0; i < 3; ++i)
            if (!(P->racer_inertia[i] > 0)) { *why = "racer_inertia must be positive"; return FPV_EPARAM; }
        if (P->racer_pid_variant > 1) { *why = "racer_pid_variant must be 0 (racer_drone_test.PID) or 1 (components.PID)"; return FPV_EINVAL; }
        if (P->racer_pid_variant == 1 && (!(P->pid_integral_clip >= 0) || !(P->pid_min_output <= P->pid_max_output)
                                          || !(P->pid_derivative_transition_rate >= 0 && P->pid_derivative_transition_rate <= 1))) {
            *why = "components.PID constants: integral_clip >= 0, min_output <= max_output, derivative_transition_rate in [0, 1]";
            return FPV_EPARAM;
        }
    }
    const double qn = sqrt(P->init_quat[0] * P->init_quat[0] + P->init_quat[1] * P->init_quat[1] +
                           P->init_quat[2] * P->init_quat[2] + P->init_quat[3] * P->init_quat[3]);
    if (!(fabs(qn - 1.0) < 1e-6)) { *why = "init_quat must be a unit quaternion"; return FPV_EPARAM; }

    memset(K, 0, sizeof(*K));
    K->dt = (float)P->dt;
    K->max_rates = (float)P->max_rates;
    K->kr = (float)P->rates_transition_rate;
    K->omkr = (float)(1.0 - P->rates_transition_rate);
    K->kt = (float)P->thrust_transition_rate;
    K->omkt = (float)(1.0 - P->thrust_transition_rate);
    const double c3 = P->thrust_poly[0], c2 = P->thrust_poly[1], c1 = P->thrust_poly[2], c0 = P->thrust_poly[3];
    const double h = 50.0;   // x = h*a + h
    K->d3 = (float)(c3 * h * h * h);
    K->d2 = (float)(3 * c3 * h * h * h + c2 * h * h);
    K->d1 = (float)(3 * c3 * h * h * h + 2 * c2 * h * h + c1 * h);
    K->d0 = (float)(c3 * h * h * h + c2 * h * h + c1 * h + c0);
    {
        const double kt = P->thrust_transition_rate;
        K->dk3 = (float)(kt * (c3 * h * h * h));
        K->dk2 = (float)(kt * (3 * c3 * h * h * h + c2 * h * h));
        K->dk1 = (float)(kt * (3 * c3 * h * h * h + 2 * c2 * h * h + c1 * h));
        K->dk0 = (float)(kt * (c3 * h * h * h + c2 * h * h + c1 * h + c0));
        K->rate_lim = (float)(P->max_rates * P->rates_transition_rate);
        K->rate_gain = -K->rate_lim;
    }
    K->inv_mass = (float)(1.0 / P->mass);
    K->g = (float)P->gravity;
    for (int i = 0; i < 3; ++i)
        K->kdrag_m[i] = (float)(0.5 * P->drag_coefficients[i] * P->air_density * P->cross_section_areas[i] / P->mass);
    K->half_k = (float)(0.5 * (M_PI / 180.0) * P->dt);
    for (int m = 0; m < 4; ++m) { K->motor_x[m] = (float)P->motor_xy[m][0]; K->motor_y[m] = (float)P->motor_xy[m][1]; }
    {   // the reference's X frame after narrowing: (c,c) (-c,c) (-c,-c) (c,-c) in any order, one |c| bit for bit
        const float c = fabsf(K->motor_x[0]);
        bool square = c > 0.0f;
        int seen = 0;
        for (int m = 0; m < 4; ++m) {
            square = square && fabsf(K->motor_x[m]) == c && fabsf(K->motor_y[m]) == c;
            seen |= 1 << ((K->motor_x[m] < 0 ? 1 : 0) | (K->motor_y[m] < 0 ? 2 : 0));
        }
        K->motor_square = (square && seen == 15) ? 1u : 0u;
        K->motor_c = c;
    }
    for (int i = 0; i < 3; ++i) { K->p0[i] = (float)P->init_position[i]; K->v0[i] = (float)P->init_velocity[i]; K->goal[i] = (float)P->goal[i]; }
    for (int i = 0; i < 4; ++i) K->q0[i] = (float)(P->init_quat[i] / qn);
    K->ceiling = (float)P->ceiling;        // +inf stays +inf
    K->r_dt = (float)P->dt;
    K->r_inv_mass = (float)(1.0 / (P->racer_mass > 0 ? P->racer_mass : 1.0));
    K->r_damp = (float)P->racer_velocity_damping;
    K->r_ang_k = P->racer_omega_dt ? (float)P->dt : 1.0f;
    K->r_ang_k_d = P->racer_omega_dt ? P->dt : 1.0;
    // omega per STEP (as written) needs the float64 rate loop; omega*dt is well conditioned in fp32
    K->r_wide = P->racer_omega_dt ? 0u : 1u;
    K->r_pid_variant = P->racer_pid_variant;
    K->rd.dt = P->dt;
    K->rd.inv_dt = 1.0 / P->dt;
    for (int i = 0; i < 3; ++i) {
        K->rd.dt_over_I[i] = P->dt / (P->racer_inertia[i] > 0 ? P->racer_inertia[i] : 1.0);
        for (int j = 0; j < 3; ++j) K->rd.gain[i][j] = P->racer_pid[i][j];
    }
    K->rd.integral_clip = P->pid_integral_clip;
    K->rd.min_output = P->pid_min_output;
    K->rd.max_output = P->pid_max_output;
    K->rd.d_rate = P->pid_derivative_transition_rate;
    K->rd.om_d_rate = 1.0 - P->pid_derivative_transition_rate;
    K->rf.dt = (float)K->rd.dt; K->rf.inv_dt = (float)K->rd.inv_dt;
    for (int i = 0; i < 3; ++i) {
        K->rf.dt_over_I[i] = (float)K->rd.dt_over_I[i];
        for (int j = 0; j < 3; ++j) K->rf.gain[i][j] = (float)K->rd.gain[i][j];
    }
    K->rf.integral_clip = (float)K->rd.integral_clip; K->rf.min_output = (float)K->rd.min_output;
    K->rf.max_output = (float)K->rd.max_output; K->rf.d_rate = (float)K->rd.d_rate; K->rf.om_d_rate = (float)K->rd.om_d_rate;
    K->motor_radius = (float)P->motor_radius;
    K->ground_k_m = (float)(P->ground_spring / P->mass);
    K->ground_c_m = (float)(P->ground_damping / P->mass);
    double arm = 0.0;
    for (int m = 0; m < 4; ++m) arm = fmax(arm, sqrt(P->motor_xy[m][0] * P->motor_xy[m][0] + P->motor_xy[m][1] * P->motor_xy[m][1]));
    K->contact_reach = (float)(arm + fmax(P->motor_radius, 0.0) + 1e-3);
    K->noise.tau = (float)P->noise_transition;
    K->noise.omtau = (float)(1.0 - P->noise_transition);
    K->noise.gain = (float)P->noise_gain;
    K->noise.seed_lo = (uint32_t)P->noise_seed; K->noise.seed_hi = (uint32_t)(P->noise_seed >> 32);
    K->noise.id_lo = (uint32_t)P->drone_id_offset; K->noise.id_hi = (uint32_t)(P->drone_id_offset >> 32);
    K->flags = P->flags;
    // |rates| <= max_rates always (clip + convex low-pass from 0), so the largest half-angle of one
    // step is known here: up to 0.03 rad two series terms are exact to fp32, up to pi/4 the five-term
    // polynomials, beyond that the angle is reduced first (fpv_sincos3)
    const double half_max = 0.5 * (M_PI / 180.0) * P->dt * P->max_rates;
    K->angle_mode = half_max <= 0.03 ? FPV_ANGLE_TINY : half_max <= 0.78 ? FPV_ANGLE_SMALL : FPV_ANGLE_REDUCED;
    return FPV_OK;
}
