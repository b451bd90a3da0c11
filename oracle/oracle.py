"""ctypes front-end of the float64 CPU restatement (oracle/fpv_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from fpyv_amd/.  Parity status: PINNED (tests/golden, produced by
oracle/gen_golden.py from the reference's own Drone.step / Racer.step).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

DRONE_STATE = 19
RACER_STATE = 23


class OracleObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("_p", C.c_int32), ("x", C.c_double), ("y", C.c_double), ("z", C.c_double),
                ("radius", C.c_double), ("height", C.c_double)]


class OracleParams(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("gravity", C.c_double), ("mass", C.c_double), ("max_rates", C.c_double),
        ("rates_transition_rate", C.c_double), ("thrust_transition_rate", C.c_double),
        ("thrust_poly", C.c_double * 4),
        ("drag_coefficients", C.c_double * 3), ("cross_section_areas", C.c_double * 3),
        ("air_density", C.c_double),
        ("motor_xy", (C.c_double * 2) * 4),
        ("racer_mass", C.c_double), ("racer_inertia", C.c_double * 3),
        ("racer_pid", (C.c_double * 3) * 3),
        ("racer_velocity_damping", C.c_double),
        ("racer_omega_dt", C.c_int32), ("ground", C.c_int32),
        ("motor_radius", C.c_double), ("ground_spring", C.c_double), ("ground_damping", C.c_double),
        ("n_objects", C.c_int32), ("_pad2", C.c_int32), ("objects", OracleObject * 8),
        ("racer_pid_variant", C.c_int32), ("_pad3", C.c_int32),
        ("pid_integral_clip", C.c_double), ("pid_min_output", C.c_double), ("pid_max_output", C.c_double),
        ("pid_derivative_transition_rate", C.c_double),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    if force or not os.path.isfile(os.path.join(_BUILD, "libfpv_oracle.so")):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


_libs = {}


def lib(native: bool = False) -> C.CDLL:
    """Portable x86-64 build (gcc -O2, no -march: it must run on whatever CPU the GPU box has), or - for
    bench.py's cpu_baseline leg only - the same source built -O3 -march=native on the host that runs it."""
    name = "libfpv_oracle_native.so" if native else "libfpv_oracle.so"
    if name not in _libs:
        if native:
            subprocess.run(["make", "-C", _HERE, "-B", "native"], check=True, stdout=subprocess.DEVNULL)
        else:
            build()
        L = C.CDLL(os.path.join(_BUILD, name))
        dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        L.fpvo_drone_step_batch.argtypes = [C.POINTER(OracleParams), C.c_int64, C.c_int, dp, dp, C.c_int,
                                            dp, dp, u8p, C.c_int]
        L.fpvo_drone_step_batch.restype = None
        L.fpvo_drone_step_guided.argtypes = [C.POINTER(OracleParams), dp, dp, dp, dp, C.c_double, dp, u8p]
        L.fpvo_drone_step_guided.restype = None
        L.fpvo_racer_step_batch.argtypes = [C.POINTER(OracleParams), C.c_int64, C.c_int, dp, dp, C.c_int, C.c_int]
        L.fpvo_racer_step_batch.restype = None
        L.fpvo_matrix_to_quat_wxyz.argtypes = [dp, dp]
        L.fpvo_quat_wxyz_to_matrix.argtypes = [dp, dp]
        L.fpvo_euler_zyx_matrix.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.fpvo_max_threads.restype = C.c_int
        L.fpvo_pid_call.argtypes = [dp, dp, C.c_double, C.c_double]
        L.fpvo_pid_call.restype = C.c_double
        _libs[name] = L
    return _libs[name]


def pack_params(p) -> OracleParams:
    """fpyv_amd.params.DroneParams (plain float64 fields) -> oracle struct."""
    o = OracleParams()
    o.dt, o.gravity, o.mass, o.max_rates = p.dt, p.gravity, p.mass, p.max_rates
    o.rates_transition_rate, o.thrust_transition_rate = p.rates_transition_rate, p.thrust_transition_rate
    o.thrust_poly[:] = [float(x) for x in p.thrust_poly]
    o.drag_coefficients[:] = [float(x) for x in p.drag_coefficients]
    o.cross_section_areas[:] = [float(x) for x in p.cross_section_areas]
    o.air_density = p.air_density
    for m in range(4):
        o.motor_xy[m][0], o.motor_xy[m][1] = float(p.motor_xy[m][0]), float(p.motor_xy[m][1])
    o.racer_mass = p.racer_mass
    o.racer_inertia[:] = [float(x) for x in p.racer_inertia]
    for i in range(3):
        for j in range(3):
            o.racer_pid[i][j] = float(p.racer_pid[i][j])
    o.racer_velocity_damping = p.racer_velocity_damping
    o.racer_omega_dt = int(bool(p.racer_omega_dt))
    o.ground = int(bool(getattr(p, "ground", False)))
    o.motor_radius, o.ground_spring, o.ground_damping = p.motor_radius, p.ground_spring, p.ground_damping
    o.racer_pid_variant = int(getattr(p, "racer_pid_variant", 0))
    o.pid_integral_clip = float(getattr(p, "pid_integral_clip", 1.0))
    o.pid_min_output = float(getattr(p, "pid_min_output", 0.3))
    o.pid_max_output = float(getattr(p, "pid_max_output", 1.0))
    o.pid_derivative_transition_rate = float(getattr(p, "pid_derivative_transition_rate", 0.5))
    objs = list(getattr(p, "objects", ()) or ())
    o.n_objects = len(objs)
    for k, ob in enumerate(objs):      # (type, x, y, z, radius, height)
        o.objects[k].type = int(ob[0])
        o.objects[k].x, o.objects[k].y, o.objects[k].z = float(ob[1]), float(ob[2]), float(ob[3])
        o.objects[k].radius, o.objects[k].height = float(ob[4]), float(ob[5])
    return o


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def euler_zyx_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    E = np.empty(9)
    lib().fpvo_euler_zyx_matrix(roll, pitch, yaw, _dp(E))
    return E.reshape(3, 3)


def matrix_to_quat(R: np.ndarray) -> np.ndarray:
    """[...,3,3] -> [...,4] (w,x,y,z), w >= 0."""
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(-1, 9)
    q = np.empty((R.shape[0], 4))
    L = lib()
    for i in range(R.shape[0]):
        L.fpvo_matrix_to_quat_wxyz(_dp(R[i]), _dp(q[i]))
    return q


def quat_to_matrix(q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 4)
    R = np.empty((q.shape[0], 9))
    L = lib()
    for i in range(q.shape[0]):
        L.fpvo_quat_wxyz_to_matrix(_dp(q[i]), _dp(R[i]))
    return R.reshape(-1, 3, 3)


def drone_initial_state(n: int, position, velocity, ypr_deg) -> np.ndarray:
    """Drone.reset (components.py:150-169): state [n,19] = p, v, R (row-major), prev_rates=0, prev_thrust=0.
    position/velocity/ypr broadcast from [3] or are given per drone [n,3]."""
    s = np.zeros((n, DRONE_STATE))
    s[:, 0:3] = np.asarray(position, dtype=np.float64)
    s[:, 3:6] = np.asarray(velocity, dtype=np.float64)
    ypr = np.broadcast_to(np.asarray(ypr_deg, dtype=np.float64), (n, 3))
    for i in range(n):
        r = np.deg2rad(ypr[i])
        s[i, 6:15] = euler_zyx_matrix(r[0], r[1], r[2]).reshape(9)   # consumed as (roll, pitch, yaw)
    return s


def drone_run(p, state: np.ndarray, actions: np.ndarray, steps: Optional[int] = None,
              wind=(0.0, 0.0, 0.0), threads: int = 1, native: bool = False
              ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Advance `state` [n,19] in place.  actions: [steps,n,4] (one batch per step) or [n,4]
    (held for `steps` steps).  Returns (state, accel [n,3], done [n]) after the last step."""
    n = state.shape[0]
    assert state.shape == (n, DRONE_STATE) and state.dtype == np.float64 and state.flags.c_contiguous
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    per_step = actions.ndim == 3
    if per_step:
        steps = actions.shape[0] if steps is None else steps
        assert actions.shape == (steps, n, 4)
    else:
        assert steps is not None and actions.shape == (n, 4)
    accel = np.zeros((n, 3))
    done = np.zeros(n, dtype=np.uint8)
    w = np.asarray(wind, dtype=np.float64)
    op = pack_params(p)
    lib(native).fpvo_drone_step_batch(C.byref(op), n, steps, _dp(state), _dp(actions), int(per_step), _dp(w),
                                    _dp(accel), done.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
    return state, accel, done


def simd_lib(native: bool = False) -> C.CDLL:
    """fpv_oracle_simd.c (Drone.step with the drones as the vector axis): the portable build for the tests, or the
    -march=native build on the host that runs bench.py's cpu_baseline.simd_across_drones leg."""
    name = "libfpv_oracle_simd_native.so" if native else "libfpv_oracle_simd.so"
    if name not in _libs:
        if native:
            subprocess.run(["make", "-C", _HERE, "-B", os.path.join("_build", name)], check=True, stdout=subprocess.DEVNULL)
        elif not os.path.isfile(os.path.join(_BUILD, name)):
            subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_BUILD, name))
        dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        L.fpvs_drone_step_batch.argtypes = [C.POINTER(OracleParams), C.c_int64, C.c_int, dp, dp, C.c_int, dp, dp, u8p, C.c_int]
        L.fpvs_drone_step_batch.restype = C.c_int
        _libs[name] = L
    return _libs[name]


def drone_run_simd(p, state: np.ndarray, actions: np.ndarray, steps: Optional[int] = None,
                   wind=(0.0, 0.0, 0.0), threads: int = 1, native: bool = False
                   ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """drone_run through the SIMD-across-drones build (same arguments, same results to rounding level)."""
    n = state.shape[0]
    assert state.shape == (n, DRONE_STATE) and state.dtype == np.float64 and state.flags.c_contiguous
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    per_step = actions.ndim == 3
    if per_step:
        steps = actions.shape[0] if steps is None else steps
        assert actions.shape == (steps, n, 4)
    else:
        assert steps is not None and actions.shape == (n, 4)
    accel = np.zeros((n, 3))
    done = np.zeros(n, dtype=np.uint8)
    w = np.asarray(wind, dtype=np.float64)
    op = pack_params(p)
    rc = simd_lib(native).fpvs_drone_step_batch(C.byref(op), n, steps, _dp(state), _dp(actions), int(per_step), _dp(w),
                                                _dp(accel), done.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
    if rc != 0:
        raise ValueError("the SIMD-across-drones build covers object_list = [] or [Ground] only")
    return state, accel, done


def drone_run_guided(p, state: np.ndarray, actions: np.ndarray, rotations, thrust_forces, wind=(0.0, 0.0, 0.0)
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One drone (state [19]), T steps with the guidance arguments of Drone.step (components.py:230-232):
    rotations [T,3,3] and thrust_forces [T]; a NaN thrust force means `rotation_matrix=None` on that step.
    Returns (states [T,19] after every step, accel [T,3], done [T])."""
    assert state.shape == (DRONE_STATE,) and state.dtype == np.float64
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    T = actions.shape[0]
    rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(T, 9)
    tf = np.asarray(thrust_forces, dtype=np.float64)
    w = np.asarray(wind, dtype=np.float64)
    op, L = pack_params(p), lib()
    states, accel, done = np.zeros((T, DRONE_STATE)), np.zeros((T, 3)), np.zeros(T, dtype=np.uint8)
    for t in range(T):
        d = C.c_uint8(0)
        L.fpvo_drone_step_guided(C.byref(op), _dp(state), _dp(actions[t]), _dp(w),
                                 None if np.isnan(tf[t]) else _dp(rot[t]), float(tf[t]), _dp(accel[t]), C.byref(d))
        states[t], done[t] = state, d.value
    return states, accel, done


def racer_initial_state(n: int) -> np.ndarray:
    """Racer.reset (racer_drone_test.py:85-93): zeros, identity quaternion (x,y,z,w), PID first-call flag set."""
    s = np.zeros((n, RACER_STATE))
    s[:, 9] = 1.0
    s[:, 19] = 1.0
    return s


def racer_run(p, state: np.ndarray, actions: np.ndarray, steps: Optional[int] = None,
              threads: int = 1) -> np.ndarray:
    n = state.shape[0]
    assert state.shape == (n, RACER_STATE) and state.dtype == np.float64 and state.flags.c_contiguous
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    per_step = actions.ndim == 3
    if per_step:
        steps = actions.shape[0] if steps is None else steps
    op = pack_params(p)
    lib().fpvo_racer_step_batch(C.byref(op), n, steps, _dp(state), _dp(actions), int(per_step), threads)
    return state


def pid_run(gains, current: np.ndarray, target: np.ndarray):
    """components.PID over a sequence: gains = (kP, kI, kD, dt, integral_clip, min_output, max_output,
    derivative_transition_rate).  Returns (out, integral, derivative, error) per call."""
    k = np.asarray(gains, dtype=np.float64)
    st = np.array([0.0, 0.0, 0.0, 1.0])
    T = len(current)
    out, integ, der, err = (np.zeros(T) for _ in range(4))
    L = lib()
    for t in range(T):
        out[t] = L.fpvo_pid_call(_dp(k), _dp(st), float(current[t]), float(target[t]))
        integ[t], der[t], err[t] = st[0], st[1], st[2]
    return out, integ, der, err


def max_threads() -> int:
    return int(lib().fpvo_max_threads())
