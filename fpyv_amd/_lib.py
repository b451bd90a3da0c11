"""ctypes binding of libfpv_hip.so (include/fpv_abi.h).

There is exactly one compute path: the HIP library.  If it is missing or does not load, importing
this module's `lib()` raises - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfpv_hip.so")

FPV_ABI_VERSION = 8
FPV_OK = 0
FPV_MODE_DRONE, FPV_MODE_RACER = 0, 1
FPV_DRONE_ROWS, FPV_RACER_ROWS = 14, 29
FPV_FLAG_AUTO_RESET = 1
FPV_FLAG_GROUND = 2
FPV_FLAG_FP16_STATE = 4
FPV_FLAG_STICK_NOISE = 8
FPV_HALF_PAIR_ROWS = 5
FPV_HALF_HALVES = 11          # binary16 values per drone in state_h (5 pair rows + 1 half row)
FPV_OBS_AOS_DIM = 16

# state rows (fpv_abi.h)
PX, PY, PZ, VX, VY, VZ, QW, QX, QY, QZ, RX, RY, RZ, THRUST = range(14)
R_OMEGA, R_IERR, R_LERR, R_FIRST, R_OMEGA_LO, R_IERR_LO, R_DFILT = 10, 13, 16, 19, 20, 23, 26

# every symbol include/fpv_abi.h declares
EXPORTS = ("fpv_abi_version", "fpv_sizeof", "fpv_state_rows", "fpv_algorithmic_bytes", "fpv_handle_algorithmic_bytes", "fpv_create", "fpv_destroy",
           "fpv_reset", "fpv_step", "fpv_rollout", "fpv_step_n", "fpv_rollout_graph", "fpv_return_triple", "fpv_widen_state", "fpv_set_params", "fpv_set_step_counter", "fpv_get_step_counter", "fpv_set_rotation", "fpv_get_rotation", "fpv_recommended_ld",
           "fpv_recommended_ld_device", "fpv_check_cache_model", "fpv_device_cache_model", "fpv_get_cache_model",
           "fpv_diag_stream_copy", "fpv_diag_stream_copy_wide", "fpv_diag_busy", "fpv_diag_xcd_map", "fpv_pid_reset", "fpv_pid_call", "fpv_comm_unique_id", "fpv_comm_create", "fpv_comm_destroy", "fpv_comm_info",
           "fpv_allgather_done", "fpv_allgather_f32", "fpv_last_error",
           "fpv_error_name", "fpv_encoding_id")


class FpvParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_uint32), ("flags", C.c_uint32), ("racer_omega_dt", C.c_uint32),
        ("dt", C.c_double), ("gravity", C.c_double), ("mass", C.c_double), ("max_rates", C.c_double),
        ("rates_transition_rate", C.c_double), ("thrust_transition_rate", C.c_double),
        ("thrust_poly", C.c_double * 4),
        ("drag_coefficients", C.c_double * 3), ("cross_section_areas", C.c_double * 3),
        ("air_density", C.c_double),
        ("motor_xy", (C.c_double * 2) * 4),
        ("init_position", C.c_double * 3), ("init_velocity", C.c_double * 3), ("init_quat", C.c_double * 4),
        ("ceiling", C.c_double), ("goal", C.c_double * 3),
        ("racer_mass", C.c_double), ("racer_inertia", C.c_double * 3), ("racer_pid", (C.c_double * 3) * 3),
        ("racer_velocity_damping", C.c_double),
        ("motor_radius", C.c_double), ("ground_spring", C.c_double), ("ground_damping", C.c_double),
        ("noise_transition", C.c_double), ("noise_gain", C.c_double),
        ("noise_seed", C.c_uint64), ("drone_id_offset", C.c_uint64),
        ("racer_pid_variant", C.c_uint32), ("_reserved0", C.c_uint32),
        ("pid_integral_clip", C.c_double), ("pid_min_output", C.c_double), ("pid_max_output", C.c_double),
        ("pid_derivative_transition_rate", C.c_double),
    ]


FPV_MAX_OBJECTS = 8
OBJ_GROUND, OBJ_CYLINDER, OBJ_SPHERE = 0, 1, 2
FPV_PID_ROWS = 4       # integral, prev_derivative, previous_error, is_first
FPV_COMM_ID_BYTES = 128


class FpvPidParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("_reserved", C.c_uint32), ("kP", C.c_double), ("kI", C.c_double),
                ("kD", C.c_double), ("dt", C.c_double), ("integral_clip", C.c_double), ("min_output", C.c_double),
                ("max_output", C.c_double), ("derivative_transition_rate", C.c_double)]


class FpvObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float),
                ("radius", C.c_float), ("height", C.c_float)]


class FpvObjects(C.Structure):
    _fields_ = [("count", C.c_int32), ("obj", FpvObject * FPV_MAX_OBJECTS)]


def pack_objects(objects) -> FpvObjects:
    """sequence of (type, x, y, z, radius, height) in object_list order -> fpv_objects_t"""
    objs = list(objects)
    if len(objs) > FPV_MAX_OBJECTS:
        raise ValueError(f"at most {FPV_MAX_OBJECTS} collision objects")
    t = FpvObjects()
    t.count = len(objs)
    for k, o in enumerate(objs):
        t.obj[k].type = int(o[0])
        t.obj[k].x, t.obj[k].y, t.obj[k].z = float(o[1]), float(o[2]), float(o[3])
        t.obj[k].radius, t.obj[k].height = float(o[4]), float(o[5])
    return t


class FpvBuffers(C.Structure):
    _fields_ = [
        ("state", C.c_void_p), ("ld", C.c_int64), ("action", C.c_void_p), ("reward", C.c_void_p),
        ("done", C.c_void_p), ("done_bits", C.c_void_p), ("accel", C.c_void_p), ("ep_return", C.c_void_p),
        ("ep_length", C.c_void_p), ("last_return", C.c_void_p), ("last_length", C.c_void_p),
        ("wind", C.c_float * 3), ("rounding_seed", C.c_uint32), ("state_h", C.c_void_p),
        ("pos_comp", C.c_void_p), ("noise_state", C.c_void_p), ("action_out", C.c_void_p), ("action_ld", C.c_int64), ("objects", C.c_void_p), ("obs_aos", C.c_void_p),
        ("done_bits_stride", C.c_int64), ("rotation_override", C.c_void_p), ("thrust_override", C.c_void_p),
        ("state_h_thrust", C.c_void_p),
    ]


def pack_params(p, auto_reset: bool = False, fp16_state: bool = False, stick_noise: bool = False,
                noise_seed: int = 0, drone_id_offset: int = 0) -> FpvParams:
    """DroneParams -> fpv_params_t."""
    s = FpvParams()
    s.struct_size = C.sizeof(FpvParams)
    s.mode = int(p.mode)
    s.flags = ((FPV_FLAG_AUTO_RESET if auto_reset else 0) | (FPV_FLAG_GROUND if getattr(p, "ground", False) else 0)
               | (FPV_FLAG_FP16_STATE if fp16_state else 0) | (FPV_FLAG_STICK_NOISE if stick_noise else 0))
    s.noise_transition = float(getattr(p, "noise_transition", 0.1))
    s.noise_gain = float(getattr(p, "noise_gain", 1.0))
    s.noise_seed, s.drone_id_offset = int(noise_seed) & (2 ** 64 - 1), int(drone_id_offset)
    s.racer_omega_dt = int(bool(p.racer_omega_dt))
    s.dt, s.gravity, s.mass, s.max_rates = float(p.dt), float(p.gravity), float(p.mass), float(p.max_rates)
    s.rates_transition_rate = float(p.rates_transition_rate)
    s.thrust_transition_rate = float(p.thrust_transition_rate)
    s.thrust_poly[:] = [float(x) for x in p.thrust_poly]
    s.drag_coefficients[:] = [float(x) for x in p.drag_coefficients]
    s.cross_section_areas[:] = [float(x) for x in p.cross_section_areas]
    s.air_density = float(p.air_density)
    for m in range(4):
        s.motor_xy[m][0], s.motor_xy[m][1] = float(p.motor_xy[m][0]), float(p.motor_xy[m][1])
    s.init_position[:] = [float(x) for x in p.init_position]
    s.init_velocity[:] = [float(x) for x in p.init_velocity]
    s.init_quat[:] = [float(x) for x in p.init_quat]
    s.ceiling = float(p.ceiling)
    s.goal[:] = [float(x) for x in p.goal]
    s.racer_mass = float(p.racer_mass)
    s.racer_inertia[:] = [float(x) for x in p.racer_inertia]
    for i in range(3):
        for j in range(3):
            s.racer_pid[i][j] = float(p.racer_pid[i][j])
    s.racer_velocity_damping = float(p.racer_velocity_damping)
    s.racer_pid_variant = int(getattr(p, "racer_pid_variant", 0))
    s.pid_integral_clip = float(getattr(p, "pid_integral_clip", 1.0))
    s.pid_min_output = float(getattr(p, "pid_min_output", 0.3))
    s.pid_max_output = float(getattr(p, "pid_max_output", 1.0))
    s.pid_derivative_transition_rate = float(getattr(p, "pid_derivative_transition_rate", 0.5))
    s.motor_radius, s.ground_spring, s.ground_damping = float(p.motor_radius), float(p.ground_spring), float(p.ground_damping)
    return s


class FpvCacheModel(C.Structure):
    """fpv_cache_model_t: what a device says about itself, held against the cache model of the rotation / row stride."""
    _fields_ = [("struct_size", C.c_uint32), ("matches", C.c_int32), ("compute_units", C.c_int32), ("xcds", C.c_int32),
                ("l2_bytes_per_xcd", C.c_int64), ("infinity_cache_bytes", C.c_int64), ("arch", C.c_char * 64), ("reason", C.c_char * 256)]

    def as_dict(self):
        return {"matches": bool(self.matches), "arch": self.arch.decode(), "compute_units": int(self.compute_units), "xcds": int(self.xcds),
                "l2_bytes_per_xcd": int(self.l2_bytes_per_xcd), "infinity_cache_bytes": int(self.infinity_cache_bytes),
                "reason": self.reason.decode() or None}


class FpvError(RuntimeError):
    def __init__(self, code: int, name: str, msg: str):
        super().__init__(f"{name} ({code}): {msg}")
        self.code, self.name = code, name


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libfpv_hip.so (once).  torch is imported first so that the HIP runtime torch ships
    (same SONAME, libamdhip64.so.7) is the one both sides use - one runtime, shared streams and
    device pointers."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  fpyv_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 / libhsa-runtime64 from torch/lib)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    vp, i64, pp, pb = C.c_void_p, C.c_int64, C.POINTER(FpvParams), C.POINTER(FpvBuffers)
    L.fpv_abi_version.restype = C.c_int
    L.fpv_state_rows.argtypes = [C.c_int]
    L.fpv_algorithmic_bytes.argtypes = [C.c_int]
    L.fpv_handle_algorithmic_bytes.argtypes = [vp]
    L.fpv_create.argtypes = [pp, i64, C.c_int, C.POINTER(vp)]
    L.fpv_destroy.argtypes = [vp]
    L.fpv_destroy.restype = None
    L.fpv_reset.argtypes = [vp, pb, vp, vp, vp, vp, vp]
    L.fpv_step.argtypes = [vp, pb, vp]
    L.fpv_rollout.argtypes = [vp, pb, C.c_int, i64, i64, vp]
    L.fpv_rollout_graph.argtypes = [vp, pb, C.c_int, i64, i64, vp]
    L.fpv_step_n.argtypes = [vp, pb, C.c_int, i64, i64, vp]
    L.fpv_return_triple.argtypes = [vp, pb, vp, vp, vp, vp]
    L.fpv_widen_state.argtypes = [vp, pb, vp, i64, vp]
    L.fpv_set_params.argtypes = [vp, pp]
    L.fpv_set_step_counter.argtypes = [vp, C.c_uint64]
    L.fpv_get_step_counter.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.fpv_set_rotation.argtypes = [vp, C.c_int64]
    L.fpv_get_rotation.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fpv_recommended_ld.argtypes = [i64]
    L.fpv_recommended_ld.restype = i64
    L.fpv_recommended_ld_device.argtypes = [i64, C.c_int]
    L.fpv_recommended_ld_device.restype = i64
    L.fpv_check_cache_model.argtypes = [C.c_char_p, C.c_int, i64, C.POINTER(FpvCacheModel)]
    L.fpv_device_cache_model.argtypes = [C.c_int, C.POINTER(FpvCacheModel)]
    L.fpv_get_cache_model.argtypes = [vp, C.POINTER(FpvCacheModel)]
    L.fpv_diag_stream_copy.argtypes = [vp, vp, i64, vp]
    L.fpv_diag_stream_copy_wide.argtypes = [vp, vp, i64, vp]
    L.fpv_diag_busy.argtypes = [C.c_double, vp]
    L.fpv_diag_xcd_map.argtypes = [vp, i64, vp]
    L.fpv_comm_unique_id.argtypes = [vp]
    L.fpv_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.fpv_comm_destroy.argtypes = [vp]
    L.fpv_comm_destroy.restype = None
    L.fpv_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fpv_allgather_done.argtypes = [vp, vp, vp, i64, vp]
    L.fpv_allgather_f32.argtypes = [vp, vp, vp, i64, vp]
    L.fpv_pid_reset.argtypes = [vp, i64, i64, vp, C.c_int, vp]
    L.fpv_pid_call.argtypes = [C.POINTER(FpvPidParams), vp, i64, i64, vp, vp, C.c_float, vp, vp, C.c_int, vp]
    L.fpv_last_error.restype = C.c_char_p
    L.fpv_error_name.argtypes = [C.c_int]
    L.fpv_error_name.restype = C.c_char_p
    L.fpv_encoding_id.argtypes = [C.c_int]
    L.fpv_encoding_id.restype = C.c_char_p
    L.fpv_sizeof.argtypes = [C.c_int]
    if L.fpv_abi_version() != FPV_ABI_VERSION:
        raise ImportError(f"libfpv_hip.so ABI {L.fpv_abi_version()} != binding {FPV_ABI_VERSION} - rebuild the library "
                          "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    for which, struct in ((0, FpvParams), (1, FpvBuffers), (2, FpvObjects), (3, FpvPidParams), (4, FpvCacheModel)):
        if L.fpv_sizeof(which) != C.sizeof(struct):
            raise ImportError(f"{struct.__name__}: ctypes declares {C.sizeof(struct)} bytes, libfpv_hip.so has "
                              f"{L.fpv_sizeof(which)} - _lib.py and include/fpv_abi.h are out of step")
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        L = lib()
        raise FpvError(rc, L.fpv_error_name(rc).decode(), L.fpv_last_error().decode())
    return rc


def state_rows(mode: int) -> int:
    return FPV_DRONE_ROWS if mode == FPV_MODE_DRONE else FPV_RACER_ROWS


def algorithmic_bytes(mode: int) -> int:
    """state read + write, action read, reward + done write (fpv_algorithmic_bytes): the SURVEY 8d
    figures, 133 B for the drone and 181 B for the 20 base rows of the Racer; a live handle reports
    what its kernel variant really moves (fpv_handle_algorithmic_bytes)."""
    return (FPV_DRONE_ROWS if mode == FPV_MODE_DRONE else 20) * 8 + 16 + 4 + 1
