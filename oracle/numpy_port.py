"""NumPy-vectorised restatement of Drone.step for N drones at once (float64) - SURVEY 8(d)(iii): the "idiomatic
Python" CPU baseline, i.e. what a reader of the reference would write first to batch it: the same formulas as
/root/reference/src/utils/components.py:179-248 and src/utils/kinematics.py:15-49 with a leading drone axis
(3x3 attitude matrices, R <- R E^T applied twice, Horner cubic in throttle percent, quirks Q1-Q4, Q6 kept).

TEST INFRASTRUCTURE ONLY, like everything under oracle/: checked against the C oracle (tests/test_numpy_port.py), timed
by bench.py's cpu_baseline leg as a second reported baseline.  Never imported by fpyv_amd.  No collision objects
(object_list = [] - the headline workload); the ground flag (components.py:235-240) is included.
"""
from __future__ import annotations

import numpy as np

DEG2RAD = np.pi / 180.0


def euler_zyx(roll, pitch, yaw):
    """[n, 3, 3] Rz(yaw) Ry(pitch) Rx(roll) - helper_functions.py:39-44."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    E = np.empty(roll.shape + (3, 3))
    E[..., 0, 0] = cy * cp; E[..., 0, 1] = cy * sp * sr - sy * cr; E[..., 0, 2] = cy * sp * cr + sy * sr
    E[..., 1, 0] = sy * cp; E[..., 1, 1] = sy * sp * sr + cy * cr; E[..., 1, 2] = sy * sp * cr - cy * sr
    E[..., 2, 0] = -sp;     E[..., 2, 1] = cp * sr;                E[..., 2, 2] = cp * cr
    return E


def initial_state(n, position, velocity, ypr_deg):
    """dict of arrays: p [n,3], v [n,3], R [n,3,3], prev_rates [n,3], prev_thrust [n] - Drone.reset, components.py:150-169."""
    ang = np.broadcast_to(np.asarray(ypr_deg, float), (n, 3)) * DEG2RAD
    return dict(p=np.broadcast_to(np.asarray(position, float), (n, 3)).copy(), v=np.broadcast_to(np.asarray(velocity, float), (n, 3)).copy(),
                R=euler_zyx(ang[:, 0], ang[:, 1], ang[:, 2]), prev_rates=np.zeros((n, 3)), prev_thrust=np.zeros(n))


def step(P, S, action, wind=(0.0, 0.0, 0.0)):
    """One Drone.step for every drone, in place on S.  action [n, 4].  Returns (R_new @ acc [n,3], done [n] bool)."""
    p, v, R = S["p"], S["v"], S["R"]
    # components.py:185-189: stick -> rate command, clipped, low-passed
    cmd = np.clip(-action[:, 0:3] * P.max_rates, -P.max_rates, P.max_rates)
    rates = cmd * P.rates_transition_rate + S["prev_rates"] * (1 - P.rates_transition_rate)
    S["prev_rates"] = rates
    # :136, :192-194: cubic in throttle percent, low-passed, unclamped
    x = 100 * (action[:, 3] + 1) / 2
    c = P.thrust_poly
    T = (((c[0] * x + c[1]) * x + c[2]) * x + c[3]) * P.thrust_transition_rate + S["prev_thrust"] * (1 - P.thrust_transition_rate)
    S["prev_thrust"] = T
    # kinematics.py:33-38 drag (wind ADDED, rho = P.air_density), :41-45 gravity, :48-49 thrust along R[:, 2]
    vs = v + np.asarray(wind, float)
    vb = np.einsum("nji,nj->ni", R, vs)                                       # R^T v_s
    fb = -0.5 * P.air_density * (np.asarray(P.drag_coefficients) * np.asarray(P.cross_section_areas)) * vb * np.linalg.norm(vs, axis=1, keepdims=True)
    F = np.einsum("nij,nj->ni", R, fb) + T[:, None] * R[:, :, 2]
    F[:, 2] -= P.gravity * P.mass
    # components.py:235-240: any motor below z = 0 on the pre-update pose (not latched)
    m = np.asarray(P.motor_xy)                                               # [4, 2], body frame, z = 0
    mz = p[:, 2:3] + R[:, 2, 0:1] * m[None, :, 0] + R[:, 2, 1:2] * m[None, :, 1]
    done = (mz < 0).any(axis=1)
    acc = F / P.mass                                                          # :242-243
    # kinematics.py:21-23 + components.py:218: p with the old v, then v, then the attitude increment TWICE
    p += v * P.dt
    v += acc * P.dt
    a = rates * DEG2RAD * P.dt
    Et = np.swapaxes(euler_zyx(a[:, 0], a[:, 1], a[:, 2]), 1, 2)
    R = R @ Et @ Et
    S["R"] = R
    return np.einsum("nij,nj->ni", R, acc), done                              # components.py:248 (third return value)


def run(P, S, actions, wind=(0.0, 0.0, 0.0)):
    """actions [T, n, 4] (or [n, 4] with steps given by the caller's loop).  Returns the last (accel, done)."""
    out = None
    for t in range(actions.shape[0]):
        out = step(P, S, actions[t], wind)
    return out


def as_oracle_rows(S):
    """[n, 19] rows in the C oracle's layout: p3 v3 R9 prev_rates3 prev_thrust."""
    n = S["p"].shape[0]
    return np.concatenate([S["p"], S["v"], S["R"].reshape(n, 9), S["prev_rates"], S["prev_thrust"][:, None]], axis=1)
