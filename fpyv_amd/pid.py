"""`PID` - the reference's guidance PID (/root/reference/src/utils/components.py:15-54) for N drones at once.

Same constructor arguments, `reset()` and `__call__(current, target)` as the reference class; every
instance steps `num_envs` independent controllers with one HIP kernel launch (fpv_pid_call).  The
reference appends to three history arrays on every call (components.py:45-51; unbounded) - that is
not reproduced; `.error`, `.integral` and `.derivative` expose the current values as device tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional

import numpy as np
import torch

from . import _lib


class PID:
    def __init__(self, kP, kI, kD, dt, integral_clip=1, min_output=0.3, max_output=1, derivative_transition_rate=0.5,
                 num_envs: int = 1, device: Any = "cuda:0"):
        self.kP, self.kI, self.kD, self.dt = float(kP), float(kI), float(kD), float(dt)
        self.integral_clip, self.min_output, self.max_output = float(integral_clip), float(min_output), float(max_output)
        self.derivative_transition_rate = float(derivative_transition_rate)
        self.n = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("fpyv_amd runs on the GPU only (device must be cuda:N); there is no CPU path")
        self._L = _lib.lib()
        self._dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ld = (self.n + 63) // 64 * 64
        self.state = torch.zeros((_lib.FPV_PID_ROWS, self.ld), dtype=torch.float32, device=self.device)
        self.output = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.error = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.reset()

    def _params(self) -> _lib.FpvPidParams:
        p = _lib.FpvPidParams()
        p.struct_size = C.sizeof(_lib.FpvPidParams)
        p.kP, p.kI, p.kD, p.dt = self.kP, self.kI, self.kD, self.dt
        p.integral_clip, p.min_output, p.max_output = self.integral_clip, self.min_output, self.max_output
        p.derivative_transition_rate = self.derivative_transition_rate
        return p

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """PID.reset (components.py:35-41); `mask` [num_envs] resets only the flagged controllers."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        _lib.check(self._L.fpv_pid_reset(self.state.data_ptr(), self.ld, self.n, m.data_ptr() if m is not None else None,
                                         self._dev, self._stream()))
        self._keep = m

    def __call__(self, current, target) -> torch.Tensor:
        """error = current - target (components.py:44); returns the clipped output, [num_envs] on the device."""
        cur = self._per_drone(current)
        tgt, tgt_ptr, tgt_scalar = None, None, 0.0
        if np.ndim(target) == 0 or (hasattr(target, "numel") and target.numel() == 1) or np.size(target) == 1:
            tgt_scalar = float(target.reshape(-1)[0]) if hasattr(target, "reshape") else float(target)
        else:                                        # one target per drone: tensor, ndarray, list
            tgt = self._per_drone(target)
            tgt_ptr = tgt.data_ptr()
        _lib.check(self._L.fpv_pid_call(C.byref(self._params()), self.state.data_ptr(), self.ld, self.n, cur.data_ptr(),
                                        tgt_ptr, tgt_scalar, self.output.data_ptr(), self.error.data_ptr(), self._dev,
                                        self._stream()))
        self._keep = (cur, tgt)
        return self.output

    def _per_drone(self, x) -> torch.Tensor:
        t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32).reshape(-1)
        if t.numel() not in (1, self.n):
            raise ValueError(f"expected a scalar or {self.n} values, got {t.numel()}")
        return t.expand(self.n).contiguous()

    @property
    def integral(self) -> torch.Tensor:
        return self.state[0, :self.n]

    @property
    def derivative(self) -> torch.Tensor:
        return self.state[1, :self.n]
