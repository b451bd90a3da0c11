"""Multi-GPU sharding: one process per GPU, drones split into contiguous shards.

Every drone is independent (nothing in Drone.step couples two drones,
/root/reference/src/utils/components.py:220-248), so the physics needs no collective.  The only
exchange is the per-step all-gather of the done mask (and, on request, episode returns) for a
learner that wants the global view; it runs through torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) on the collective's own stream, double-buffered so the
next step's kernel never waits for it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n_total % world_size ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_total: int, world_size: int) -> List[int]:
    return [shard_range(n_total, world_size, r)[1] - shard_range(n_total, world_size, r)[0]
            for r in range(world_size)]


def pack_done_bits(done: torch.Tensor) -> torch.Tensor:
    """uint8 [n] -> int64 [ceil(n/64)], bit (i % 64) of word i // 64 = done[i]; the layout the
    kernel's wave ballot writes (fpv_buffers_t.done_bits).  Host/torch helper for tests and for
    backends without the kernel."""
    n = done.numel()
    pad = (-n) % 64
    d = torch.cat([done.to(torch.int64), done.new_zeros(pad, dtype=torch.int64)]) if pad else done.to(torch.int64)
    w = d.view(-1, 64)
    lo = (w[:, :63] << torch.arange(63, dtype=torch.int64, device=done.device)).sum(dim=1)
    hi = w[:, 63] * torch.iinfo(torch.int64).min     # bit 63 is the sign bit
    return lo + hi


def unpack_done_bits(bits: torch.Tensor, n: int) -> torch.Tensor:
    sh = torch.arange(64, dtype=torch.int64, device=bits.device)
    return ((bits.view(-1, 1) >> sh) & 1).reshape(-1)[:n].to(torch.uint8)


class DoneGather:
    """Double-buffered asynchronous all-gather of one per-shard tensor per step.

    usage per step t:   buf = g.slot(t)           # tensor the kernel writes this step
                        ... launch the step with done/done_bits -> buf ...
                        g.launch(t)               # enqueue all_gather(buf) (async)
                        g.result(t - 1)           # optional: last step's global mask [world, shard]
    slot(t) first waits (stream-side, not host-side) for the all-gather that last read that buffer.
    Shards must be the same size on every rank (pad the last one)."""

    def __init__(self, shard_shape, dtype, device, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.local = [torch.zeros(shard_shape, dtype=dtype, device=device) for _ in range(2)]
        # flat concatenation: the one output shape both RCCL and gloo accept for all_gather_into_tensor
        self._flat = [torch.zeros(self.world * self.local[0].numel(), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [f.view((self.world,) + tuple(shard_shape)) for f in self._flat]
        self.pending: List[Optional[object]] = [None, None]

    def slot(self, t: int) -> torch.Tensor:
        k = t & 1
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        return self.local[k]

    def launch(self, t: int) -> None:
        k = t & 1
        self.pending[k] = self._dist.all_gather_into_tensor(self._flat[k], self.local[k].view(-1), group=self.group,
                                                            async_op=True)

    def result(self, t: int) -> torch.Tensor:
        k = t & 1
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        return self.gathered[k]

    def drain(self) -> None:
        for k in (0, 1):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None
