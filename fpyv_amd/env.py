"""Batched host API over the HIP stepper.

`DroneBatch` mirrors the reference's `Drone` (/root/reference/src/utils/components.py:72-248) for N
drones at once - same method names, argument meaning and return triple:

    Drone.reset(position, velocity, ypr)                       components.py:150-169
    Drone.step(action, wind_velocity_vector, object_list)      components.py:220-248
        -> (rotation_matrix.T, angular_velocity_matrix, rotation_matrix @ acceleration)
    Drone.position / .velocity / .done                         components.py:171-177, :104

`RacerBatch` does the same for `Racer` (/root/reference/tests/racer_drone_test.py:68-103), and
`FpvVecEnv` is the gym-style `reset() -> obs`, `step(a) -> (obs, reward, done, info)` surface the
reference's env scripts use (tests/rotation_pid.py:57-78).

All state lives in torch tensors on the GPU; a step is one ctypes call that enqueues one kernel on
torch's current stream.  Nothing here computes physics on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .params import DroneParams, MODE_DRONE, MODE_RACER, load_params


# identifiers stored in checkpoints (state_dict): what the fp16 storage words and the in-kernel stick-noise stream mean.
# The strings are defined next to the code they describe (csrc/fpv_math.h FPV_STATE_H_ENCODING_ID / FPV_NOISE_GENERATOR_ID) and
# exported as fpv_encoding_id(0 / 1); tests/test_abi_load.py holds the two copies below to them, so they cannot drift apart.
STATE_H_ENCODING = "abi5: v f16+5-bit low words, q smallest-three 15-bit fixed point, rates/thrust f16"     # csrc/fpv_math.h fpv_pack_half
NOISE_GENERATOR = "abi5: philox4x32-7, table-driven inverse normal CDF"                                     # csrc/fpv_math.h fpv_stick_noise
# the first ABI whose fp16 checkpoints carry `state_h_encoding`; ABI 5 wrote the same encoding without the label
_FIRST_ABI_WITH_ENCODING_LABEL = 6
CHECKPOINT_LAYOUT = "columns"        # row tensors are stored as their logical columns [rows, num_envs]: independent of the row stride


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _Batch:
    """Owns the SoA state tensor and the C handle of one shard of drones on one GPU."""

    def __init__(self, params: DroneParams, num_envs: int, device: Any = "cuda:0", auto_reset: bool = False,
                 track_episodes: bool = False, with_accel: bool = False, with_done_bits: bool = False,
                 fp16_state: bool = False, rounding_seed: int = 0, with_obs_aos: bool = False,
                 stick_noise: bool = False, noise_seed: int = 0, drone_id_offset: int = 0,
                 with_action_out: bool = False, kahan_position: bool = False):
        if num_envs <= 0:
            raise ValueError("num_envs must be positive")
        self.params = params
        self.n = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("fpyv_amd runs on the GPU only (device must be cuda:N); there is no CPU path")
        self._L = _lib.lib()
        self.mode = int(params.mode)
        self.rows = _lib.state_rows(self.mode)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev_index = int(dev_index)
        # the row stride for THIS device: the L2-aware rule on the MI355X it was measured on, the model-free one elsewhere
        self.ld = int(_lib.check(self._L.fpv_recommended_ld_device(self.n, self._dev_index)))
        self._handle = C.c_void_p()
        self.fp16_state = bool(fp16_state)
        self.rounding_seed = int(rounding_seed) & 0xFFFFFFFF
        self.stick_noise = bool(stick_noise)
        self._pack_kw = dict(fp16_state=self.fp16_state, stick_noise=self.stick_noise, noise_seed=noise_seed,
                             drone_id_offset=drone_id_offset)
        self._cparams = _lib.pack_params(params, auto_reset=auto_reset, **self._pack_kw)
        _lib.check(self._L.fpv_create(C.byref(self._cparams), self.n, dev_index, C.byref(self._handle)))
        f32 = dict(dtype=torch.float32, device=self.device)
        if self.fp16_state:
            # BASELINE config 4: position rows fp32; the rest as eleven 16-bit words: five rows of word pairs
            # (vx,vy) (vz,v_low) (qa,qb) (qc,rx) (ry,rz) and one row of thrust halves: 89 B per env-step
            # (v, rates, thrust binary16 - v with 5-bit low words -, q smallest-three fixed point: csrc/fpv_math.h)
            self.state = torch.zeros((3, self.ld), **f32)
            self.state_h = torch.zeros(_lib.FPV_HALF_HALVES * self.ld, dtype=torch.float16, device=self.device)
        else:
            self.state = torch.zeros((self.rows, self.ld), **f32)
            self.state_h = None
        self.reward = torch.zeros(self.n, **f32)
        # the kernel writes one byte, exactly 0 or 1, per drone: a torch.bool tensor is that byte array, so `done` is
        # the kernel's own output (done[N] bool) and `done_u8` the same memory seen as uint8
        self.done = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        self.done_u8 = self.done.view(torch.uint8)
        self.accel = torch.zeros((3, self.ld), **f32) if with_accel else None
        self.done_bits = (torch.zeros(_round_up(self.n, 64) // 64, dtype=torch.int64, device=self.device)
                          if with_done_bits else None)
        if track_episodes:
            self.ep_return = torch.zeros(self.n, **f32)
            self.ep_length = torch.zeros(self.n, dtype=torch.int32, device=self.device)
            self.last_return = torch.zeros(self.n, **f32)
            self.last_length = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        else:
            self.ep_return = self.ep_length = self.last_return = self.last_length = None
        # Kahan compensation rows of the position accumulation (10^4+-step fp32 accuracy, config 1)
        self.pos_comp = torch.zeros((6, self.ld), **f32) if kahan_position else None
        # in-kernel EMA stick noise state (x_s per channel) and the action actually applied
        self.noise_state = torch.zeros((4, self.ld), **f32) if self.stick_noise else None
        self.action_out = torch.zeros((self.n, 4), **f32) if with_action_out else None
        # optional row-major [num_envs, 16] observation written by the kernel through an LDS transpose
        self.obs_aos = (torch.zeros((self.n, _lib.FPV_OBS_AOS_DIM), **f32) if with_obs_aos else None)
        self._bcast_action = None
        self._objects = None            # the bound fpv_objects_t (None = no collision world bound)
        self._object_rows = None
        self._last_action, self._last_action_ptr = None, 0
        self._override_keep = None
        self._done_bits_keep = None
        self._ashape = torch.Size((self.n, 4))
        self._steps_launched = 0        # mirrors the handle's launch counter (fpv_set_step_counter)
        self._buf = _lib.FpvBuffers()
        self._buf_ref = C.byref(self._buf)
        self._fpv_step = self._L.fpv_step
        self._fill_buffers()

    # -- plumbing ---------------------------------------------------------------------------------
    def _fill_buffers(self) -> None:
        b = self._buf
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        b.state, b.ld = self.state.data_ptr(), self.ld
        b.reward, b.done = self.reward.data_ptr(), self.done_u8.data_ptr()
        b.done_bits, b.accel = ptr(self.done_bits), ptr(self.accel)
        b.ep_return, b.ep_length = ptr(self.ep_return), ptr(self.ep_length)
        b.last_return, b.last_length = ptr(self.last_return), ptr(self.last_length)
        b.wind[0] = b.wind[1] = b.wind[2] = 0.0
        b.state_h, b.rounding_seed = ptr(self.state_h), self.rounding_seed
        b.obs_aos = ptr(self.obs_aos)
        b.noise_state, b.action_out = ptr(self.noise_state), ptr(self.action_out)
        b.pos_comp = ptr(self.pos_comp)

    def rows_f32(self, r0: int, r1: int) -> torch.Tensor:
        """[num_envs, r1-r0] fp32 values of state rows r0..r1-1 (fpv_abi.h row numbering), whatever the
        storage format; a zero-copy view for fp32 storage, a converted copy for fp16 rows."""
        if not self.fp16_state:
            return self.state[r0:r1, :self.n].t()
        if r1 <= 3:
            return self.state[r0:r1, :self.n].t()
        # one launch (fpv_widen_state) into a fresh [14, ld] tensor: a copy, like before, valid for as long as the caller keeps it
        wide = torch.empty((_lib.FPV_DRONE_ROWS, self.ld), dtype=torch.float32, device=self.device)
        _lib.check(self._L.fpv_widen_state(self._handle, self._buf_ref, wide.data_ptr(), self.ld, self._stream()))
        return wide[r0:r1, :self.n].t()

    def storage_words(self) -> torch.Tensor:
        """[11, ld] int16: the eleven 16-bit storage words of every drone of an fp16-state batch in storage order - vx vy vz
        (binary16), v_low (three 5-bit low words), qa qb qc (smallest-three 15-bit fixed point + index bits), rx ry rz
        thrust (binary16) - a copy assembled from the five pair rows and the thrust row of `state_h`.  The decoded fp32
        values are `rows_f32` (one launch of fpv_widen_state)."""
        ld, npair = self.ld, _lib.FPV_HALF_PAIR_ROWS
        raw = self.state_h.view(torch.int16)
        pairs = raw[:2 * npair * ld].view(npair, ld, 2).permute(0, 2, 1).reshape(2 * npair, ld)
        return torch.cat([pairs, raw[2 * npair * ld:].view(1, ld)], dim=0)

    def algorithmic_bytes(self) -> int:
        return int(self._L.fpv_handle_algorithmic_bytes(self._handle))

    def _stream(self) -> int:
        """hipStream_t of torch's current stream on this device (raw handle; ~5x cheaper than building
        a torch.cuda.Stream object on every step)."""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index)
        except AttributeError:                       # private fast path gone in some future torch
            return torch.cuda.current_stream(self.device).cuda_stream

    def _action_ptr(self, action: Any) -> Optional[int]:
        if action is None:
            if not self.stick_noise:
                raise ValueError("action=None is only meaningful with stick_noise=True (pure noise sticks)")
            return None
        # hottest path: the very tensor object of the previous step (a policy that writes its output in place), still at
        # the same address: everything below was checked then
        if action is self._last_action and action.data_ptr() == self._last_action_ptr and action.shape == self._ashape:
            self._keepalive = action                 # a rollout in between may have replaced what `throttle` reports
            return self._last_action_ptr
        self._last_action = None
        # hot path: a contiguous float32 [num_envs, 4] tensor on the env's device
        if (type(action) is torch.Tensor and action.dtype is torch.float32 and action.shape == self._ashape
                and action.is_contiguous() and action.device == self.state.device):
            self._keepalive = action
            self._buf.action_ld = 0
            self._last_action, self._last_action_ptr = action, action.data_ptr()
            return self._last_action_ptr
        # SoA sticks [4, num_envs] (e.g. the output of `W @ obs_soa`): consumed in place, no transpose
        if (type(action) is torch.Tensor and action.dim() == 2 and action.shape[0] == 4 and action.shape[1] == self.n
                and self.n != 4 and action.dtype is torch.float32 and action.stride(1) == 1 and action.stride(0) >= self.n
                and action.device == self.state.device):
            self._keepalive = action
            self._buf.action_ld = action.stride(0)
            return action.data_ptr()
        self._buf.action_ld = 0
        action = self._coerce_action(action)
        self._keepalive = action
        return action.data_ptr()

    def _coerce_action(self, action: Any) -> torch.Tensor:
        """Whatever the reference's callers pass as sticks - a list, a NumPy array, a [4] broadcast, a tensor of another
        dtype / device / stride - as a contiguous float32 [num_envs, 4] tensor on the env's device (a copy only if needed)."""
        if not torch.is_tensor(action):
            action = torch.as_tensor(np.asarray(action, dtype=np.float32), device=self.device)
        if action.dim() == 1:
            if action.numel() != 4:
                raise ValueError("action must be [4] or [num_envs, 4]")
            if self._bcast_action is None:
                self._bcast_action = torch.empty((self.n, 4), dtype=torch.float32, device=self.device)
            self._bcast_action.copy_(action.to(device=self.device, dtype=torch.float32).expand(self.n, 4))
            action = self._bcast_action
        if action.shape != (self.n, 4):
            raise ValueError(f"action must have shape ({self.n}, 4), got {tuple(action.shape)}")
        if action.dtype in (torch.float16, torch.bfloat16, torch.float64) and action.device == self.state.device:
            action = self._cast_sticks(action)           # one rule for step(), rollout() and step_async()
        if action.dtype != torch.float32 or action.device != self.state.device or not action.is_contiguous():
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        return action

    _warned_cast = False

    def _cast_sticks(self, action: torch.Tensor) -> torch.Tensor:
        """A stick tensor of another floating dtype as contiguous float32 (one cast kernel per call; warned about once: a
        half-precision policy that cares writes `.float()` into a preallocated buffer itself).  The same rule in step(),
        rollout() and step_async()."""
        if not _Batch._warned_cast:
            import warnings
            warnings.warn(f"sticks of dtype {action.dtype} are cast to float32 on every call (the kernels read float32 sticks since ABI 6)",
                          RuntimeWarning, stacklevel=3)
            _Batch._warned_cast = True
        return action.to(torch.float32).contiguous()

    def set_step_counter(self, step: int) -> None:
        """64-bit step index keying the stick-noise stream / stochastic rounding (counts the steps launched, from 0)."""
        if not 0 <= int(step) < 2 ** 64:
            raise ValueError("the step counter is an unsigned 64-bit integer")
        _lib.check(self._L.fpv_set_step_counter(self._handle, int(step)))
        self._steps_launched = int(step)

    def step_counter(self) -> int:
        """The handle's own 64-bit step counter (fpv_get_step_counter)."""
        v = C.c_uint64()
        _lib.check(self._L.fpv_get_step_counter(self._handle, C.byref(v)))
        return int(v.value)

    def set_rotation(self, drones: int = -1) -> None:
        """How far the start of the traversal moves back from launch to launch (fpv_set_rotation; every single-step kernel:
        drone fp32 / fp16 state / AoS head / Racer): -1 automatic - two cache tiers: the L2s' share of drones when a launch
        writes more than the eight L2s hold (2^20 drones: 2^19), the Infinity Cache's share beyond 256 MiB (2^23 drones: 2^22),
        plain order below, and plain order on a device that is not the one the model was measured on (`cache_model`) -,
        0 the plain order, > 0 that many drones.  Results do not depend on it.  What it buys is a property of step-only
        chains (profiles/r06_closed_loop.md: with a policy kernel between steps it is neutral).  A hipGraph replay
        (rollout(graph=True)) counts its own rotation from its first node and leaves this one where it was."""
        _lib.check(self._L.fpv_set_rotation(self._handle, int(drones)))

    @property
    def rotation(self) -> int:
        """Drones the start moves back per launch (0 = plain order): what fpv_get_rotation reports."""
        v = C.c_int64()
        _lib.check(self._L.fpv_get_rotation(self._handle, C.byref(v)))
        return int(v.value)

    @property
    def cache_model(self) -> Dict[str, Any]:
        """What fpv_create found when it held the device against the cache model of the rotation and the row stride
        (fpv_get_cache_model): `matches`, the device's architecture / compute units / L2 size, and - when it does not match -
        the reason the handle runs the plain order."""
        m = _lib.FpvCacheModel()
        _lib.check(self._L.fpv_get_cache_model(self._handle, C.byref(m)))
        return m.as_dict()

    def set_params(self, params: DroneParams, auto_reset: Optional[bool] = None) -> None:
        flags_auto = bool(self._cparams.flags & _lib.FPV_FLAG_AUTO_RESET) if auto_reset is None else auto_reset
        cp = _lib.pack_params(params, auto_reset=flags_auto, **self._pack_kw)
        _lib.check(self._L.fpv_set_params(self._handle, C.byref(cp)))
        self.params, self._cparams = params, cp

    # -- checkpoint / resume (the reference has none; state is just tensors here) ------------------
    _CKPT_TENSORS = ("state", "state_h", "reward", "done", "ep_return", "ep_length", "last_return",
                     "last_length", "noise_state", "pos_comp")
    _CKPT_ROW_TENSORS = ("state", "noise_state", "pos_comp")        # [rows, ld]: stored as their logical columns [rows, num_envs]

    def _state_h_views(self, t: Optional[torch.Tensor] = None, ld: Optional[int] = None):
        """(pair rows [5, ld, 2], thrust row [ld]) int16 views of an fp16 storage tensor laid out with row stride `ld`"""
        t = self.state_h if t is None else t
        ld = self.ld if ld is None else ld
        raw, npair = t.view(torch.int16), _lib.FPV_HALF_PAIR_ROWS
        return raw[:2 * npair * ld].view(npair, ld, 2), raw[2 * npair * ld:2 * npair * ld + ld]

    def state_dict(self) -> Dict[str, Any]:
        """Everything needed to continue a run bit-for-bit: the device tensors (cloned) and the step counter that keys the
        stick-noise stream / stochastic rounding.  Row tensors are stored as their LOGICAL columns - `state` [rows, num_envs],
        the fp16 words as `state_h` [11, num_envs] int16 in storage order (five pair rows interleaved, then thrust) - so a
        checkpoint does not depend on the row stride the writing library chose (fpv_recommended_ld has changed between
        rounds and differs between devices)."""
        d: Dict[str, Any] = {}
        for k in self._CKPT_TENSORS:
            t = getattr(self, k, None)
            if t is None:
                continue
            if k in self._CKPT_ROW_TENSORS:
                d[k] = t[:, :self.n].clone()
            elif k == "state_h":
                pairs, thrust = self._state_h_views()
                d[k] = torch.cat([pairs[:, :self.n].permute(0, 2, 1).reshape(-1, self.n), thrust[:self.n].view(1, self.n)], dim=0)
            else:
                d[k] = t.clone()
        d["step_counter"] = int(self._steps_launched)
        d["num_envs"], d["mode"] = self.n, self.mode
        d["layout"], d["ld"] = CHECKPOINT_LAYOUT, self.ld                # ld: for the record only - load does not need it
        # what the bits mean: the fp16 storage words and the stick-noise streams changed between ABI versions
        d["abi_version"] = _lib.FPV_ABI_VERSION
        if self.fp16_state:
            d["state_h_encoding"] = STATE_H_ENCODING
        if self.stick_noise:
            d["noise_generator"] = NOISE_GENERATOR
        return d

    def load_state_dict(self, d: Dict[str, Any]) -> None:
        """Accepts this library's checkpoints (logical columns) and those of rounds <= 5 (padded tensors with the writer's own
        row stride, whatever it was: the stride is read off the tensor's shape)."""
        import warnings
        if d["num_envs"] != self.n or d["mode"] != self.mode:
            raise ValueError("checkpoint was taken from a batch of different size or mode")
        if "done_u8" in d and "done" not in d:        # checkpoints written before the bool view existed
            d = dict(d, done=d["done_u8"].bool())
        if self.fp16_state and d.get("state_h_encoding") != STATE_H_ENCODING:
            abi = d.get("abi_version")
            if "state_h_encoding" not in d and (abi is None or abi < _FIRST_ABI_WITH_ENCODING_LABEL) and "state_h" in d \
                    and d["state_h"].numel() % _lib.FPV_HALF_HALVES == 0:
                # an ABI-5 checkpoint: the encoding this library still reads (ABI 6 only made non-unit quaternion fields saturate
                # instead of wrapping - the stored bits of unit quaternions are the same), written before checkpoints were labelled.
                # (ABI <= 4 stored eleven separate half rows; such a file decodes as garbage and cannot be told apart by its shape.)
                warnings.warn("fp16-state checkpoint without a `state_h_encoding` label (written by an ABI-5 library): read as "
                              f"{STATE_H_ENCODING!r}; a checkpoint of ABI <= 4 must be widened with the library that wrote it", RuntimeWarning, stacklevel=2)
            else:
                # another library's eleven words decode as garbage here (e.g. its qw half would be read as the v_low bits)
                raise ValueError(f"fp16-state checkpoint with storage encoding {d.get('state_h_encoding')!r} (ABI {abi if abi is not None else '<= 5, unrecorded'}); "
                                 f"this library reads {STATE_H_ENCODING!r} - widen the old state with the library that wrote it and load the fp32 rows")
        if self.stick_noise and d.get("noise_generator") != NOISE_GENERATOR:
            warnings.warn(f"checkpoint was written with stick-noise generator {d.get('noise_generator')!r}, this library runs {NOISE_GENERATOR!r}: "
                          "the run continues with a different (equally distributed) stick stream, not bit for bit", RuntimeWarning, stacklevel=2)
        for k in self._CKPT_TENSORS:
            if k not in d:
                continue
            mine, src = getattr(self, k, None), d[k]
            if mine is None:
                raise ValueError(f"checkpoint has {k!r} but this batch was built without it")
            if k in self._CKPT_ROW_TENSORS:
                if src.dim() != 2 or src.shape[0] != mine.shape[0] or src.shape[1] < self.n:
                    raise ValueError(f"checkpoint tensor {k!r} has shape {tuple(src.shape)}; expected [{mine.shape[0]}, >= {self.n}]")
                mine[:, :self.n].copy_(src[:, :self.n])             # logical columns, or a padded tensor of any row stride
            elif k == "state_h":
                pairs, thrust = self._state_h_views()
                npair = _lib.FPV_HALF_PAIR_ROWS
                if src.dim() == 2:                                   # [11, num_envs] int16 in storage order
                    if tuple(src.shape) != (_lib.FPV_HALF_HALVES, self.n):
                        raise ValueError(f"checkpoint tensor 'state_h' has shape {tuple(src.shape)}; expected [{_lib.FPV_HALF_HALVES}, {self.n}]")
                    w = src.view(torch.int16)
                    pairs[:, :self.n].copy_(w[:2 * npair].view(npair, 2, self.n).permute(0, 2, 1))
                    thrust[:self.n].copy_(w[2 * npair])
                else:                                                # rounds <= 5: the flat padded tensor, FPV_HALF_HALVES * ld_then halves
                    ld_then = src.numel() // _lib.FPV_HALF_HALVES
                    if src.numel() % _lib.FPV_HALF_HALVES or ld_then < self.n:
                        raise ValueError(f"checkpoint tensor 'state_h' has {src.numel()} halves: not {_lib.FPV_HALF_HALVES} rows of >= {self.n}")
                    op, ot = self._state_h_views(src, ld_then)
                    pairs[:, :self.n].copy_(op[:, :self.n])
                    thrust[:self.n].copy_(ot[:self.n])
            else:
                mine.copy_(src)
        self.set_step_counter(d["step_counter"])

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._L.fpv_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw stepping -----------------------------------------------------------------------------
    def _reset_raw(self, mask=None, position=None, velocity=None, ypr=None) -> None:
        def dev3(x):
            if x is None:
                return None
            t = torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x,
                                dtype=torch.float32, device=self.device)
            return t.expand(self.n, 3).contiguous()
        pos, vel, ang = dev3(position), dev3(velocity), dev3(ypr)
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.n,):
                raise ValueError("mask must have shape (num_envs,)")
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        _lib.check(self._L.fpv_reset(self._handle, C.byref(self._buf), ptr(m), ptr(pos), ptr(vel), ptr(ang),
                                     self._stream()))
        self._keepalive_reset = (pos, vel, ang, m)

    def set_objects(self, object_list) -> None:
        """The collision world of the following `rollout` calls: the same `object_list` a `step` takes, bound until it
        is replaced.  (`step` binds its own argument on every call, like the reference's Drone.step does - an empty
        default there clears what was bound here.)  The table is read on the host when a launch is enqueued: objects
        that move (Target) are re-bound by the caller after each update, exactly as the reference re-passes the list."""
        self._set_objects(object_list)

    def _set_objects(self, object_list) -> None:
        """Bind the step's object_list (host-side table, read by fpv_step during the call)."""
        if object_list is None or not len(object_list):
            if self._buf.objects:
                self._objects = None
                self._buf.objects = None
            return
        from .objects import to_rows
        rows = to_rows(object_list)
        if rows:
            if rows != self._object_rows or self._objects is None:      # a world that did not move is not re-packed
                self._objects = _lib.pack_objects(rows)
                self._object_rows = rows
            self._buf.objects = C.addressof(self._objects)
        else:
            self._objects = None
            self._buf.objects = None

    def _step_raw(self, action: Any, wind: Optional[Sequence[float]] = None) -> None:
        b = self._buf
        b.action = self._action_ptr(action)
        if wind is not None:
            b.wind[0], b.wind[1], b.wind[2] = float(wind[0]), float(wind[1]), float(wind[2])
        rc = self._fpv_step(self._handle, self._buf_ref, self._stream())
        if rc < 0:
            _lib.check(rc)
        self._steps_launched = (self._steps_launched + 1) & 0xFFFFFFFFFFFFFFFF

    def set_done_bits_target(self, target: Any = None, stride_words: int = 0) -> None:
        """Where the kernel writes the bit-packed done mask (one wave ballot per 64 drones):
        an int64 tensor of at least ceil(num_envs / 64) words or a raw device address; None restores the
        batch's own `done_bits` tensor (or switches the mask off when the batch was built without one).
        `stride_words` > 0 makes a k-step `rollout` write step t's mask at target + t * stride_words words
        (e.g. the rows of a [k, words] bucket that a collective ships afterwards); 0 = each step overwrites."""
        words = _round_up(self.n, 64) // 64
        if target is None:
            self._done_bits_keep = None
            self._buf.done_bits = self.done_bits.data_ptr() if self.done_bits is not None else None
        elif torch.is_tensor(target):
            if target.dtype != torch.int64 or not target.is_contiguous() or target.device != self.state.device:
                raise ValueError("done-bits target must be a contiguous int64 tensor on the env's device")
            if target.numel() < words:
                raise ValueError(f"done-bits target needs at least {words} words")
            self._done_bits_keep = target
            self._buf.done_bits = target.data_ptr()
        else:
            self._done_bits_keep = None
            self._buf.done_bits = int(target)
        if stride_words and stride_words < words:
            raise ValueError(f"stride_words must be 0 or >= {words}")
        self._buf.done_bits_stride = int(stride_words)

    def rollout(self, actions: Optional[torch.Tensor], wind: Optional[Sequence[float]] = None,
                rewards: Optional[torch.Tensor] = None, dones: Optional[torch.Tensor] = None,
                steps: Optional[int] = None, graph: Optional[bool] = None, fused: Optional[bool] = None,
                object_list: Any = None) -> None:
        """k steps without returning to Python: actions [k, num_envs, 4] (one batch per step) or
        [num_envs, 4] with `rewards`/`dones` of shape [k, num_envs] (or `steps`) giving k; actions=None
        with stick_noise=True runs `steps` steps of pure in-kernel noise sticks.  `object_list` (what `step`
        takes) is the collision world of all k steps; None keeps whatever `set_objects` / the last `step` bound.

        fused (default): ONE launch of the k-step kernel (fpv_step_n) - the drone stays in registers for
        the k steps; results are bit-identical to k single steps.  fused=False issues k single-step
        launches from one C call (fpv_rollout); graph=True replays those launches from a hipGraph cached
        in the handle (fpv_rollout_graph)."""
        b = self._buf
        if object_list is not None:
            self._set_objects(object_list)
        if actions is None:
            if not self.stick_noise or steps is None:
                raise ValueError("actions=None needs stick_noise=True and steps=k")
            k, stride = int(steps), 0
        elif actions.dim() == 3:
            k, stride = actions.shape[0], self.n * 4
            if actions.shape[1:] != (self.n, 4):
                raise ValueError(f"actions must be [k, {self.n}, 4]")
        else:
            if rewards is None and dones is None and steps is None:
                raise ValueError("held-action rollouts need rewards/dones [k, num_envs] or steps=k to define k")
            k, stride = (int(steps) if steps is not None else (rewards if rewards is not None else dones).shape[0]), 0
        if actions is not None and actions.dtype in (torch.float16, torch.bfloat16, torch.float64) and actions.device == self.state.device:
            actions = self._cast_sticks(actions)           # as step() does: the kernels read float32 sticks (ABI 6 dropped the binary16 rows)
        if actions is not None and (actions.dtype != torch.float32 or not actions.is_contiguous()
                                    or actions.device != self.state.device):
            raise ValueError("actions must be a contiguous float32 tensor on the env's device (float16 / bfloat16 / float64 "
                             "tensors there are cast once per call)")
        b.action = actions.data_ptr() if actions is not None else None
        b.action_ld = 0
        if wind is not None:
            b.wind[0], b.wind[1], b.wind[2] = float(wind[0]), float(wind[1]), float(wind[2])
        out_stride = 0
        saved = (b.reward, b.done)
        if rewards is not None or dones is not None:
            out_stride = self.n
            for t, dts in ((rewards, (torch.float32,)), (dones, (torch.uint8, torch.bool))):
                if t is not None and (t.shape != (k, self.n) or t.dtype not in dts or not t.is_contiguous()):
                    raise ValueError("rewards must be float32 [k, n], dones uint8 or bool [k, n], contiguous")
            b.reward = rewards.data_ptr() if rewards is not None else None
            b.done = dones.data_ptr() if dones is not None else None
        try:
            if graph:
                fn = self._L.fpv_rollout_graph
            elif fused is None or fused:
                if self.obs_aos is not None:
                    if fused:
                        raise ValueError("the fused k-step kernel does not write obs_aos rows; use fused=False")
                    fn = self._L.fpv_rollout
                else:
                    fn = self._L.fpv_step_n
            else:
                fn = self._L.fpv_rollout
            _lib.check(fn(self._handle, C.byref(b), int(k), stride, out_stride, self._stream()))
            self._steps_launched = (self._steps_launched + int(k)) & 0xFFFFFFFFFFFFFFFF
            self._keepalive = actions               # what `throttle` reports: the last step's sticks
            self._last_action = None                # the next step() re-validates its tensor (b.action / action_ld were rewritten here)
        finally:
            b.reward, b.done = saved

    # -- views ------------------------------------------------------------------------------------
    @property
    def position(self) -> torch.Tensor:
        """[num_envs, 3] view of the state (Drone.position, components.py:171-173)."""
        return self.rows_f32(_lib.PX, _lib.PZ + 1)

    @property
    def velocity(self) -> torch.Tensor:
        return self.rows_f32(_lib.VX, _lib.VZ + 1)

    @property
    def quaternion(self) -> torch.Tensor:
        """[num_envs, 4] (w, x, y, z), body -> world."""
        return self.rows_f32(_lib.QW, _lib.QZ + 1)

    @property
    def rotation_matrix(self) -> torch.Tensor:
        """[num_envs, 3, 3] body -> world, computed from the quaternion (helper_functions.py:100-117): for fp32 drone
        state by the return-triple kernel (one launch; the transpose of its R.T output is a view), otherwise by tensor
        operations."""
        if self.mode == MODE_DRONE and not self.fp16_state:
            rt = torch.empty((self.n, 3, 3), dtype=torch.float32, device=self.device)
            gy = torch.empty((self.n, 3, 3), dtype=torch.float32, device=self.device)
            _lib.check(self._L.fpv_return_triple(self._handle, self._buf_ref, rt.data_ptr(), gy.data_ptr(), None, self._stream()))
            return rt.transpose(-1, -2)
        return quat_to_matrix(self.quaternion)


class _Partition:
    """Columns [lo, hi) of a parent batch as a stepper of their own: the SAME device tensors (every pointer is the
    parent's, moved by `lo` elements; the row stride is the parent's), its own C handle (n = hi - lo drones, global
    ids continuing the parent's: `drone_id_offset + lo`), its own stream.  A partition's steps form an independent
    chain of kernels; chains of different partitions overlap on the GPU, which hides a part of each other's per-launch
    floor.  `lo` is a multiple of 128 - whole workgroups, whole done-mask words, 16-byte aligned rows."""

    def __init__(self, parent: "_Batch", lo: int, hi: int, stream: Optional[torch.cuda.Stream]):
        if lo % 128 or not lo < hi <= parent.n:
            raise ValueError("a partition starts at a multiple of 128 drones and is not empty")
        self.parent, self.lo, self.hi, self.n = parent, lo, hi, hi - lo
        self.stream = stream if stream is not None else torch.cuda.Stream(device=parent.device)
        self._stream_ptr = self.stream.cuda_stream
        self._L = parent._L
        kw = dict(parent._pack_kw, drone_id_offset=int(parent._pack_kw.get("drone_id_offset", 0)) + lo)
        auto = bool(parent._cparams.flags & _lib.FPV_FLAG_AUTO_RESET)
        self._cparams = _lib.pack_params(parent.params, auto_reset=auto, **kw)
        self._handle = C.c_void_p()
        _lib.check(self._L.fpv_create(C.byref(self._cparams), self.n, parent._dev_index, C.byref(self._handle)))
        self._buf = _lib.FpvBuffers()
        self._buf_ref = C.byref(self._buf)
        self._ashape, self._keep = torch.Size((self.n, 4)), None
        self.steps_launched = 0
        self.rebind()

    def rebind(self) -> None:
        """Point this partition's fpv_buffers_t at the parent's tensors (again, after the parent re-bound something)."""
        pb, b, lo = self.parent._buf, self._buf, self.lo
        off = lambda base, elem_bytes, per_drone=1: (base + lo * elem_bytes * per_drone) if base else None  # noqa: E731
        b.state, b.ld = off(pb.state, 4), pb.ld
        b.reward, b.done = off(pb.reward, 4), off(pb.done, 1)
        b.done_bits = (pb.done_bits + (lo // 64) * 8) if pb.done_bits else None
        b.accel, b.pos_comp, b.noise_state = off(pb.accel, 4), off(pb.pos_comp, 4), off(pb.noise_state, 4)
        b.ep_return, b.ep_length = off(pb.ep_return, 4), off(pb.ep_length, 4)
        b.last_return, b.last_length = off(pb.last_return, 4), off(pb.last_length, 4)
        b.action_out, b.obs_aos = off(pb.action_out, 4, 4), off(pb.obs_aos, 4, _lib.FPV_OBS_AOS_DIM)
        b.wind[0], b.wind[1], b.wind[2] = pb.wind[0], pb.wind[1], pb.wind[2]
        b.rounding_seed, b.objects = pb.rounding_seed, pb.objects
        if pb.state_h:
            # fp16 storage: the pair rows move by lo words, the row of thrust halves (two drones per word) by lo halves -
            # it gets its own pointer (fpv_buffers_t.state_h_thrust)
            b.state_h = pb.state_h + 4 * lo
            b.state_h_thrust = pb.state_h + 2 * (2 * _lib.FPV_HALF_PAIR_ROWS * pb.ld + lo)
        else:
            b.state_h = b.state_h_thrust = None
        b.done_bits_stride = 0

    def action_ptr(self, action: Any) -> Optional[int]:
        if action is None:
            if not self.parent.stick_noise:
                raise ValueError("action=None is only meaningful with stick_noise=True (pure noise sticks)")
            return None
        st = self.parent.state
        if type(action) is torch.Tensor and action.dtype in (torch.float16, torch.bfloat16, torch.float64) and action.device == st.device:
            action = self.parent._cast_sticks(action)      # as step() and rollout() do
        if type(action) is torch.Tensor and action.dtype is torch.float32 and action.device == st.device:
            if action.shape == self._ashape and action.is_contiguous():               # [n_p, 4] rows (a row slice of [N, 4] is one)
                self._buf.action_ld = 0
                self._keep = action
                return action.data_ptr()
            if (action.dim() == 2 and action.shape[0] == 4 and action.shape[1] == self.n and self.n != 4
                    and action.stride(1) == 1 and action.stride(0) >= self.n):         # SoA [4, n_p] (a column slice of [4, N] is one)
                self._buf.action_ld = action.stride(0)
                self._keep = action
                return action.data_ptr()
        raise ValueError(f"a partition's action is a float32 tensor on the env's device, [{self.n}, 4] contiguous rows or "
                         f"[4, {self.n}] with unit column stride (slices of a full-size tensor qualify)")

    def launch(self, action: Any) -> None:
        b = self._buf
        b.action = self.action_ptr(action)
        rc = self._L.fpv_step(self._handle, self._buf_ref, self._stream_ptr)
        if rc < 0:
            _lib.check(rc)
        self.steps_launched += 1

    def set_step_counter(self, step: int) -> None:
        _lib.check(self._L.fpv_set_step_counter(self._handle, int(step)))
        self.steps_launched = int(step)

    def set_params(self) -> None:
        """Take over the parent's current parameters (FpvVecEnv.set_params), keeping this partition's drone-id offset."""
        parent = self.parent
        kw = dict(parent._pack_kw, drone_id_offset=int(parent._pack_kw.get("drone_id_offset", 0)) + self.lo)
        auto = bool(parent._cparams.flags & _lib.FPV_FLAG_AUTO_RESET)
        cp = _lib.pack_params(parent.params, auto_reset=auto, **kw)
        _lib.check(self._L.fpv_set_params(self._handle, C.byref(cp)))
        self._cparams = cp

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._L.fpv_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_bounds(n: int, parts: int) -> Sequence[Tuple[int, int]]:
    """`parts` contiguous column ranges of n drones, every start a multiple of 128 (whole workgroups and mask words),
    sizes as equal as that allows; fewer ranges than asked for when n is too small to give each one a workgroup."""
    if parts < 1:
        raise ValueError("partitions must be >= 1")
    blocks = (n + 127) // 128
    parts = max(1, min(parts, blocks))
    cuts = [min(n, (blocks * k // parts) * 128) for k in range(parts)] + [n]
    return [(cuts[k], cuts[k + 1]) for k in range(parts)]


def as_drone_params(params: Any, mode: int, default_fps: Optional[float] = None) -> DroneParams:
    """DroneParams | params.yaml-shaped dict (what the reference passes to Drone(), components.py:73) |
    path of such a YAML | None (packaged defaults) -> DroneParams in the requested mode."""
    if params is None:
        p = load_params(fps=default_fps)
    elif isinstance(params, DroneParams):
        p = params
    elif isinstance(params, dict):
        from .params import params_from_dict
        p = params_from_dict(params)
    elif isinstance(params, (str, bytes)) or hasattr(params, "__fspath__"):
        p = load_params(params)
    else:
        raise TypeError("params must be a DroneParams, a params.yaml-shaped dict, a YAML path or None")
    return p if p.mode == mode else p.replace(mode=mode)


def quat_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) [...,4] -> rotation matrices [...,3,3]; a view helper for API parity, not on the
    step path (/root/reference/src/utils/helper_functions.py:100-117)."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
        2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
        2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y], dim=-1).reshape(q.shape[:-1] + (3, 3))


def euler_zyx_matrix(angles: torch.Tensor) -> torch.Tensor:
    """Rz(yaw) Ry(pitch) Rx(roll) for angles [...,3] = (roll, pitch, yaw) in radians
    (/root/reference/src/utils/helper_functions.py:39-44)."""
    r, p, y = angles.unbind(-1)
    cr, sr, cp, sp, cy, sy = r.cos(), r.sin(), p.cos(), p.sin(), y.cos(), y.sin()
    return torch.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                        -sp, cp * sr, cp * cr], dim=-1).reshape(angles.shape[:-1] + (3, 3))


def matrix_to_euler_zyx(R: torch.Tensor) -> torch.Tensor:
    """Rotation matrices [...,3,3] -> (roll, pitch, yaw) [...,3] in radians, the inverse of `euler_zyx_matrix`:
    helper_functions.rotation_matrix_to_euler_angles (/root/reference/src/utils/helper_functions.py:47-62) as it
    executes - its gimbal-lock branch is unreachable (`R[2,0] != 1 or R[2,0] != -1` is always true), so this is its
    general branch for every input."""
    x = torch.atan2(R[..., 2, 1], R[..., 2, 2])
    y = torch.asin((-R[..., 2, 0]).clamp(-1.0, 1.0))
    z = torch.atan2(R[..., 1, 0], R[..., 0, 0])
    return torch.stack([x, y, z], dim=-1)


def matrix_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Rotation matrices [...,3,3] -> unit quaternions (w,x,y,z) [...,4] with w >= 0.  Same rotation as
    helper_functions.rotation_matrix_to_quaternion (helper_functions.py:65-80), which divides by 4 qw and so loses
    its accuracy near half-turn attitudes; this one picks the largest of the four pivots (Shepperd), like the kernel's
    fpv_quat_from_rot."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
    k = t.argmax(dim=-1)
    r = torch.sqrt(t.gather(-1, k.unsqueeze(-1)).squeeze(-1).clamp_min(0)) * 2          # 4 * |largest component|
    a, b, c = R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]
    d, e, f = R[..., 0, 1] + R[..., 1, 0], R[..., 0, 2] + R[..., 2, 0], R[..., 1, 2] + R[..., 2, 1]
    q0 = torch.stack([r / 4, a / r, b / r, c / r], dim=-1)
    q1 = torch.stack([a / r, r / 4, d / r, e / r], dim=-1)
    q2 = torch.stack([b / r, d / r, r / 4, f / r], dim=-1)
    q3 = torch.stack([c / r, e / r, f / r, r / 4], dim=-1)
    kk = k.unsqueeze(-1)
    q = torch.where(kk == 0, q0, torch.where(kk == 1, q1, torch.where(kk == 2, q2, q3)))
    return torch.where(q[..., :1] < 0, -q, q)


class DroneBatch(_Batch):
    """N reference `Drone`s stepped by one HIP kernel (mode "drone")."""

    def __init__(self, params: Any = None, num_envs: int = 1, device: Any = "cuda:0", **kw):
        """`params` is what the reference's `Drone(params)` takes (components.py:73) - the nested dict read
        from config/params.yaml (it is NOT modified; the reference mutates it, :143-144) - or the path of
        such a YAML file, or a ready DroneParams; None loads the packaged defaults.  Sections the stepper
        does not use (camera, point_and_shoot, joystick paths, ...) are ignored."""
        params = as_drone_params(params, mode=MODE_DRONE)
        kw.setdefault("with_accel", True)
        super().__init__(params, num_envs, device, **kw)
        # the scalar attributes callers of the reference's Drone read (components.py:86-142)
        self.dt = params.dt
        self.max_rates = params.max_rates                      # :87
        self.mass, self.gravity = params.mass, params.gravity  # :92, :97
        self.throttle2thrust = params.thrust_from_stick        # :136  stick in [-1, 1] -> total thrust [N]
        self.thrust2throttle = params.stick_from_thrust        # :137  thrust [N] -> stick, clipped to [-1, 1]
        self.min_throttle_in_force = params.min_throttle_in_force   # :140
        self.max_throttle_in_force = params.max_throttle_in_force   # :142
        self._force_multiplier_pid = None

    @property
    def force_multiplier_pid(self):
        """Drone.force_multiplier_pid (components.py:143-145): the guidance PID built from the params' `drone.
        force_multiplier_pid` gains with min_output / max_output REPLACED by the 5 %-throttle and full-throttle forces,
        dt = 1/fps; one controller per drone (fpyv_amd.pid.PID, HIP kernel fpv_pid_kernel).  Created on first use -
        a fresh controller is in its reset state - and reset by reset() like the reference does (:166)."""
        if self._force_multiplier_pid is None:
            from .pid import PID
            kw = dict(self.params.force_multiplier_pid)
            kw["min_output"], kw["max_output"] = self.min_throttle_in_force, self.max_throttle_in_force   # :143-144
            self._force_multiplier_pid = PID(**kw, dt=self.dt, num_envs=self.n, device=self.device)         # :145
        return self._force_multiplier_pid

    def reset(self, position=None, velocity=None, ypr=None, mask=None) -> None:
        """Drone.reset: `ypr` is consumed as (roll, pitch, yaw) in degrees, like the reference
        (components.py:150-154).  Arguments broadcast from [3] or are per drone [num_envs, 3];
        None uses params.init_*."""
        self._reset_raw(mask=mask, position=position, velocity=velocity, ypr=ypr)
        if self._force_multiplier_pid is not None:
            self._force_multiplier_pid.reset(mask)                  # components.py:166

    def step(self, action, wind_velocity_vector=None, object_list=(), rotation_matrix=None, thrust_force=None,
             return_imu: bool = True):
        """Drone.step for every drone.  `object_list` holds up to 8 analytic collision objects
        (fpyv_amd.objects.Ground / Cylinder / Target, or raw (type, x, y, z, radius, height) rows) in
        the reference's list order.

        `rotation_matrix` ([3,3] or [num_envs,3,3], body -> world) with `thrust_force` (scalar or [num_envs],
        newtons) is the guidance call of simulator.py:110 (components.py:230-232): the sticks still advance
        prev_rates / prev_thrust, then the attitude is REPLACED by the matrix and the thrust is
        thrust_force * R[:,2].  A NaN thrust_force entry leaves that drone un-overridden.  Like the
        reference, `thrust_force` without `rotation_matrix` is ignored."""
        if action is None and not self.stick_noise:
            raise ValueError("action=None reads a physical joystick in the reference; pass stick values")
        if rotation_matrix is None and self._objects is None and not object_list:
            # the plain call - no guidance matrix, no collision world now or bound before: nothing to bind or to clear
            # (small batches are host-bound: this path is ~1 us shorter per step)
            self._step_raw(action, wind_velocity_vector)
        else:
            try:
                self._set_override(rotation_matrix, thrust_force)
                self._set_objects(object_list)
                self._step_raw(action, wind_velocity_vector)
            finally:
                # whatever raised (a bad object row, too many objects, a bad action): the next plain step() must not
                # inherit this call's guidance matrix
                self._buf.rotation_override = self._buf.thrust_override = None
                self._override_keep = None
        if not return_imu:
            return None
        if not self.fp16_state:
            # one small kernel (fpv_return_triple) instead of ~30 tensor operations: fresh tensors every call, like the
            # reference's fresh arrays (a caller may keep them across steps)
            rt = torch.empty((self.n, 3, 3), dtype=torch.float32, device=self.device)
            gyro = torch.empty((self.n, 3, 3), dtype=torch.float32, device=self.device)
            acc = torch.empty((self.n, 3), dtype=torch.float32, device=self.device) if self.accel is not None else None
            _lib.check(self._L.fpv_return_triple(self._handle, self._buf_ref, rt.data_ptr(), gyro.data_ptr(),
                                                 acc.data_ptr() if acc is not None else None, self._stream()))
            return rt, gyro, acc
        R = self.rotation_matrix
        rates = self.rows_f32(_lib.RX, _lib.RZ + 1)
        gyro = euler_zyx_matrix(rates)            # deg/s values used as radians, as the reference does (:247)
        acc = self.accel[:, :self.n].t() if self.accel is not None else None
        return R.transpose(-1, -2), gyro, acc

    @property
    def euler_angles(self) -> torch.Tensor:
        """[num_envs, 3] (roll, pitch, yaw) in radians of the current attitude (helper_functions.py:47-62)."""
        return matrix_to_euler_zyx(self.rotation_matrix)

    def get_gravity_force_in_drone_ref_frame(self) -> torch.Tensor:
        """Drone.get_gravity_force_in_drone_ref_frame (components.py:254-255), [num_envs, 3]: the reference multiplies
        the gravity vector by R (body -> world), not by its transpose, and hard-wires g = 9.81 whatever
        simulator.gravity says - reproduced as written."""
        g = torch.tensor([0.0, 0.0, -9.81 * self.mass], dtype=torch.float32, device=self.device)   # kinematics.py:41-45
        return self.rotation_matrix @ g

    def _set_override(self, rotation_matrix, thrust_force) -> None:
        if rotation_matrix is None:
            return                                   # components.py:230: thrust_force alone changes nothing
        if thrust_force is None:
            raise TypeError("rotation_matrix= needs thrust_force= (kinematics.thrust_vector(None, R) raises in the reference)")
        f32 = dict(dtype=torch.float32, device=self.device)
        R = torch.as_tensor(np.asarray(rotation_matrix, dtype=np.float32) if not torch.is_tensor(rotation_matrix)
                            else rotation_matrix, **f32)
        if R.shape == (3, 3):
            R = R.expand(self.n, 3, 3)
        if R.shape != (self.n, 3, 3):
            raise ValueError(f"rotation_matrix must be [3, 3] or [{self.n}, 3, 3], got {tuple(R.shape)}")
        f = torch.as_tensor(np.asarray(thrust_force, dtype=np.float32) if not torch.is_tensor(thrust_force)
                            else thrust_force, **f32).reshape(-1)
        if f.numel() == 1:
            f = f.expand(self.n)
        if f.shape != (self.n,):
            raise ValueError(f"thrust_force must be a scalar or [{self.n}], got {tuple(f.shape)}")
        self._override_keep = (R.reshape(self.n, 9).contiguous(), f.contiguous())
        self._buf.rotation_override = self._override_keep[0].data_ptr()
        self._buf.thrust_override = self._override_keep[1].data_ptr()

    @property
    def prev_rates(self) -> torch.Tensor:
        return self.rows_f32(_lib.RX, _lib.RZ + 1)

    @property
    def prev_thrust(self) -> torch.Tensor:
        return self.rows_f32(_lib.THRUST, _lib.THRUST + 1)[:, 0]

    @property
    def throttle(self) -> Optional[torch.Tensor]:
        """[num_envs] throttle stick of the last step (Drone.throttle, components.py:186; simulator.py:161 prints it).
        None before the first step or after a step driven purely by in-kernel stick noise (read `action_out` then)."""
        a = getattr(self, "_keepalive", None)
        if a is None or self._buf.action is None:
            return None
        if a.dim() == 3:                             # a rollout's [k, num_envs, 4] batch: its last step
            return a[-1, :, 3]
        return a[3] if self._buf.action_ld else a[:, 3]


class RacerBatch(_Batch):
    """N reference `Racer`s (rate PID -> torque -> omega -> attitude), mode "racer"."""

    def __init__(self, params: Any = None, num_envs: int = 1, device: Any = "cuda:0", **kw):
        params = as_drone_params(params, mode=MODE_RACER, default_fps=1000)
        super().__init__(params, num_envs, device, **kw)

    def reset(self, mask=None) -> None:
        self._reset_raw(mask=mask)

    def step(self, action, return_imu: bool = False) -> None:
        """action [num_envs, 4] = desired body rates (3) + thrust force (racer_drone_test.py:95-100).
        Racer.step returns nothing; `return_imu` exists only so a loop written for DroneBatch runs unchanged."""
        self._step_raw(action)

    @property
    def angular_velocity(self) -> torch.Tensor:
        return self.state[_lib.R_OMEGA:_lib.R_OMEGA + 3, :self.n].t()


class FpvVecEnv:
    """Gym-style vector env: reset() -> obs, step(action) -> (obs, reward, done, info).

    obs is a zero-copy [num_envs, 13] view of the SoA state (p3, v3, q4 wxyz, rates3): the kernel's
    state store IS the observation write.  reward = -|p - goal| (the reference defines none for
    `Drone`); done = ground contact (reference) or |z| > ceiling (build); with auto_reset the lane is
    re-initialised in the same kernel and obs already shows the fresh episode.

    **Split phase** (`partitions=P`, gym's VectorEnv step_async / step_wait per partition): the population is cut into P
    contiguous column ranges of the SAME tensors, each stepped by its own chain of kernels on its own stream:

        env = FpvVecEnv(params, num_envs=N, partitions=2)
        env.reset()
        for t in range(T):
            for part in range(env.partitions):
                obs, reward, done, info = env.step_wait(part)     # views of this partition's drones, after its last step
                env.step_async(part, policy(obs))                 # returns at once; the other partition's step is in flight

    Calls on the whole population (reset, load_state_dict, state_dict, set_done_bits_target, close) are ordered after
    every partition's enqueued steps on the device - a reset right after a step_async does not race it.

    While the policy looks at partition A, partition B steps: the policy of one half is hidden behind the step of the other
    (measured: + 16...26 % steps per second with a linear policy, + 16 % with a 13-64-4 MLP, profiles/r06_exp_closed_loop_split_phase.log).
    **Partitions are for loops with a policy between steps.**  A step-only loop (pre-generated or in-kernel sticks) is FASTER
    unpartitioned - 20.1 against 22.0 us per step at 2^20 drones (profiles/r06_bench_n1_step_partitions2.json): two chains side by
    side share the L2s that the single chain's rotated traversal has to itself - and faster still through `rollout()`.
    Drones keep their GLOBAL ids, so every buffer is bit-identical to the unpartitioned env's after the same number of steps,
    whatever P is.  `step()` still advances all drones (step_async + step_wait over all partitions).
    """

    def __init__(self, params: Optional[DroneParams] = None, num_envs: int = 1, device: Any = "cuda:0",
                 mode: str = "drone", auto_reset: bool = True, track_episodes: bool = True,
                 wind: Sequence[float] = (0.0, 0.0, 0.0), object_list=(), partitions: int = 1, **batch_options: Any):
        """`batch_options` go to DroneBatch / RacerBatch (stick_noise=, noise_seed=, drone_id_offset=,
        fp16_state=, with_obs_aos=, kahan_position=, with_done_bits=, ...); `object_list` is the
        collision world of every step (fpyv_amd.objects); `partitions` > 1 enables step_async / step_wait."""
        if mode not in ("drone", "racer"):
            raise ValueError(f'mode must be "drone" or "racer", got {mode!r}')
        params = params if params is not None else load_params(fps=1000)
        cls = DroneBatch if mode == "drone" else RacerBatch
        kw: Dict[str, Any] = dict(auto_reset=auto_reset, track_episodes=track_episodes)
        if mode == "drone":
            kw["with_accel"] = False
        kw.update(batch_options)
        self.batch = cls(params, num_envs, device, **kw)
        self.num_envs = self.batch.n
        self.wind = tuple(float(w) for w in wind)
        self.object_list = list(object_list)
        self.obs_dim = 13
        self.action_dim = 4
        self._obs_view = None
        self._parts: list = []
        self.stream_report: Optional[Dict[str, Any]] = None
        if int(partitions) > 1:
            self.batch._buf.wind[0], self.batch._buf.wind[1], self.batch._buf.wind[2] = self.wind
            bounds = partition_bounds(self.num_envs, int(partitions))
            if len(bounds) > 1:
                # streams whose kernel chains really overlap - with each other and with the caller's stream (where a policy
                # runs): measured, not assumed (fpyv_amd/streams.py; ~10 ms, once)
                from .streams import overlapping_streams
                with torch.cuda.device(self.batch.device):
                    streams, self.stream_report = overlapping_streams(self.batch.device, len(bounds),
                                                                      avoid=[torch.cuda.current_stream(self.batch.device)])
                self._parts = [_Partition(self.batch, lo, hi, st) for (lo, hi), st in zip(bounds, streams)]
                # (each partition's handle picks its own rotation of the traversal from ITS size; two chains side by side share the L2s,
                # and giving each half of the population's share measured no different: 21.9 us per step either way against 20.1 us
                # for the single chain at 2^20 drones - split phase is for closed loops, where the policy is what gets hidden)
                self._part_views = [None] * len(self._parts)
        self.partitions = max(1, len(self._parts))

    @property
    def obs(self) -> torch.Tensor:
        if self._obs_view is None:
            v = self.batch.rows_f32(0, 13)
            if self.batch.fp16_state:
                return v                     # a converted copy: made afresh every time
            self._obs_view = v               # fp32 storage: a view of the state tensor, which never moves
        return self._obs_view

    # -- ordering between the caller's stream and the partitions' chains --------------------------------------------------
    # Every call that touches the WHOLE population on the caller's stream (reset, load_state_dict, state_dict, close) first
    # makes that stream wait for every partition's chain - a step_async that nobody has step_wait-ed for yet is complete
    # before the reset kernel / the copies run - and afterwards makes every partition's stream wait for the caller's, so
    # the next step_async sees the result.  (gym's vector envs raise on a reset while a step is pending; here the streams
    # order it: `step_async(k, a); reset(mask)` is exactly `step(a); reset(mask)` of the single batch, bit for bit.)  The
    # waits are device-side (two event records and waits per partition, no host synchronisation).
    def _caller_waits_for_partitions(self) -> "torch.cuda.Stream":
        cur = torch.cuda.current_stream(self.batch.device)
        for P in self._parts:
            if cur != P.stream:
                cur.wait_stream(P.stream)
        return cur

    def _partitions_wait_for(self, cur: "torch.cuda.Stream") -> None:
        for P in self._parts:
            if cur != P.stream:
                P.stream.wait_stream(cur)

    def reset(self, mask=None) -> torch.Tensor:
        if not self._parts:
            self.batch.reset(mask=mask)
            return self.obs
        cur = self._caller_waits_for_partitions()      # steps still in flight on the partitions' streams finish first
        self.batch.reset(mask=mask)                    # (the step counters run on, as the unpartitioned batch's does across a reset)
        self._partitions_wait_for(cur)
        return self.obs

    def step(self, action) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict[str, Any]]:
        if self._parts:
            soa = False
            if action is not None:
                soa = (type(action) is torch.Tensor and action.dim() == 2 and action.shape[0] == 4 and action.shape[1] == self.num_envs != 4
                       and action.dtype is torch.float32 and action.stride(1) == 1 and action.device == self.batch.state.device)
                if not soa:
                    action = self.batch._coerce_action(action)       # lists, arrays, [4] broadcasts, other dtypes: as the single batch takes them
            for k, P in enumerate(self._parts):
                self.step_async(k, None if action is None else action[:, P.lo:P.hi] if soa else action[P.lo:P.hi])
            for k in range(len(self._parts)):
                self.step_wait(k)
            return self.obs, self.batch.reward, self.batch.done, self._info(self.batch, 0, self.num_envs)
        if self.object_list or self.batch._buf.objects:
            self.batch._set_objects(self.object_list)
        self.batch._step_raw(action, self.wind)
        return self.obs, self.batch.reward, self.batch.done, self._info(self.batch, 0, self.num_envs)

    @staticmethod
    def _info(batch, lo: int, hi: int) -> Dict[str, Any]:
        info: Dict[str, Any] = {}
        if batch.last_return is not None:
            whole = lo == 0 and hi == batch.n
            info["episode_return"] = batch.last_return if whole else batch.last_return[lo:hi]
            info["episode_length"] = batch.last_length if whole else batch.last_length[lo:hi]
        return info

    # -- split phase ------------------------------------------------------------------------------
    def partition_range(self, part: int) -> Tuple[int, int]:
        """[lo, hi): the drones of partition `part` (columns of every per-drone tensor of the env)."""
        P = self._part(part)
        return P.lo, P.hi

    def stream(self, part: int) -> torch.cuda.Stream:
        """The stream partition `part` steps on.  A policy run under `with torch.cuda.stream(env.stream(part))` is
        ordered with that partition's steps by the stream itself: step_async / step_wait then add no cross-stream waits."""
        return self._part(part).stream

    def _part(self, part: int) -> _Partition:
        if not self._parts:
            raise RuntimeError("this env was built with partitions=1: use step(), or build it with partitions=2")
        return self._parts[part]

    def step_async(self, part: int, action, ready: bool = False) -> None:
        """Enqueue one step of partition `part` on its own stream and return at once.  `action`: this partition's sticks,
        [n_p, 4] rows or [4, n_p] SoA (slices of full-size tensors qualify).  The step is ordered after whatever the
        caller's current stream has enqueued so far (the policy that produced `action`), unless that IS the partition's
        stream or `ready=True` says the tensor is already complete (pre-generated sticks)."""
        P = self._part(part)
        if not ready:
            cur = torch.cuda.current_stream(self.batch.device)
            if cur != P.stream:
                P.stream.wait_stream(cur)
        if self.object_list or self.batch._buf.objects:
            self.batch._set_objects(self.object_list)
            P._buf.objects = self.batch._buf.objects
        w = self.wind                                   # read on every step, like the single batch's step(action, self.wind)
        P._buf.wind[0], P._buf.wind[1], P._buf.wind[2] = w[0], w[1], w[2]
        P.launch(action)

    def step_wait(self, part: int, sync: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict[str, Any]]:
        """(obs, reward, done, info) of partition `part` - views of its columns - ordered after its last enqueued step:
        the caller's current stream waits for the partition's stream (on the device, not on the host).  `sync=False`
        skips that wait for a caller who works on the partition's own stream."""
        P = self._part(part)
        if sync:
            cur = torch.cuda.current_stream(self.batch.device)
            if cur != P.stream:
                cur.wait_stream(P.stream)
        b = self.batch
        if b.fp16_state:
            # fp16 storage: a decoded copy of this partition's columns (one launch of fpv_widen_state on ITS handle), fresh every time
            wide = torch.empty((_lib.FPV_DRONE_ROWS, _round_up(P.n, 64)), dtype=torch.float32, device=b.device)
            _lib.check(b._L.fpv_widen_state(P._handle, P._buf_ref, wide.data_ptr(), wide.shape[1], b._stream()))
            return wide[:13, :P.n].t(), b.reward[P.lo:P.hi], b.done[P.lo:P.hi], self._info(b, P.lo, P.hi)
        v = self._part_views[part]
        if v is None:
            v = self._part_views[part] = (b.state[:13, P.lo:P.hi].t(), b.reward[P.lo:P.hi], b.done[P.lo:P.hi])
        return v[0], v[1], v[2], self._info(b, P.lo, P.hi)

    def set_done_bits_target(self, target: Any = None) -> None:
        """Where the kernels write the bit-packed done mask (DroneBatch.set_done_bits_target); with partitions every
        partition writes its own words of the same row (a partition starts at a multiple of 64 drones).  Host-side only: a
        launch carries its target in its kernel arguments, so steps already enqueued still write the OLD target; the caller's
        stream is ordered after them, so what it reads from the old target next is complete."""
        self._caller_waits_for_partitions()
        self.batch.set_done_bits_target(target)
        for P in self._parts:
            P.rebind()

    def set_params(self, params: DroneParams, auto_reset: Optional[bool] = None) -> None:
        """New physics constants for the steps enqueued from now on - the batch's handle AND every partition's (each keeps
        its own global drone ids).  Host-side only: a launch carries its constants in its kernel arguments, steps already
        enqueued keep the ones they were launched with, exactly as on the single batch."""
        self.batch.set_params(params, auto_reset)
        for P in self._parts:
            P.set_params()

    def state_dict(self) -> Dict[str, Any]:
        """The batch's checkpoint; with partitions the step counters of all of them (they key the stick-noise streams).
        The clones are ordered after every partition's last enqueued step."""
        self._caller_waits_for_partitions()
        d = self.batch.state_dict()
        if self._parts:
            d["partition_step_counters"] = [P.steps_launched for P in self._parts]
            d["step_counter"] = min(d["partition_step_counters"])        # what an unpartitioned env continues from
        return d

    def load_state_dict(self, d: Dict[str, Any]) -> None:
        if not self._parts:
            self.batch.load_state_dict(d)
            return
        ctr = d.get("partition_step_counters") or [d["step_counter"]] * len(self._parts)
        if len(ctr) != len(self._parts):
            raise ValueError("checkpoint was taken with a different number of partitions")
        cur = self._caller_waits_for_partitions()      # a step still in flight must not land on top of the loaded state
        self.batch.load_state_dict(d)
        for P, c in zip(self._parts, ctr):
            P.set_step_counter(c)
        self._partitions_wait_for(cur)

    def close(self) -> None:
        for P in self._parts:                          # a handle is destroyed only after its chain has drained
            if getattr(P, "_handle", None) is not None and P._handle.value:
                P.stream.synchronize()
            P.close()
        self.batch.close()

    def __del__(self):
        # an env dropped without close(): its tensors go back to the allocator, which knows only the stream they were
        # allocated on - a partition's chain still running on ITS stream must be through before that memory can be handed out again
        try:
            for P in getattr(self, "_parts", []):
                P.stream.synchronize()
        except Exception:
            pass
