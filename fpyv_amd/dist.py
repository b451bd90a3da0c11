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
    """Double-buffered, bucketed, asynchronous all-gather of a per-shard tensor written every step.

    The kernel writes step t's mask into row t % block of the current bucket; every `block` steps the
    whole bucket goes out as ONE collective (async), so the host pays one collective launch per
    `block` steps instead of one per step (measured: a per-step `all_gather_into_tensor` costs ~25 us
    of host time, more than the 23 us step kernel; bucketed it disappears) and xGMI sees few, large
    messages.  Two buckets alternate: a bucket is only re-used after its previous collective has
    been waited for on the stream (not on the host).

        ptr = g.row(t)        # tensor the kernel writes at step t
        ... launch the step ...
        g.step_done(t)        # enqueues the all-gather when step t closed a bucket
        g.flush(t_last)       # gather a partially filled last bucket: only its filled rows travel
        g.result(b)           # [world, rows, *shard] masks of bucket b (rows = block, or the filled rows
                              # of a flushed bucket); waits for it

    Shards must be the same size on every rank (pad the last one)."""

    def __init__(self, shard_shape, dtype, device, block: int = 1, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.block = int(block)
        self.world = dist.get_world_size(group)
        shape = (self.block,) + tuple(shard_shape)
        self.local = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(2)]
        # flat concatenation: the one output shape both RCCL and gloo accept for all_gather_into_tensor
        self._flat = [torch.zeros(self.world * self.local[0].numel(), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [f.view((self.world,) + shape) for f in self._flat]
        self._view = list(self.gathered)           # what result() returns per bucket (a flushed bucket is shorter)
        self._row_numel = self.local[0][0].numel()
        self.pending: List[Optional[object]] = [None, None]
        self.launched = 0                          # collectives enqueued so far (warm-up included)
        self._row_ptrs = [[self.local[k][r].data_ptr() for r in range(self.block)] for k in range(2)]

    def row_ptr(self, t: int) -> int:
        """Device address of row(t) without building a tensor view (per-step hot path)."""
        b, r = divmod(t, self.block)
        if r == 0:
            self._wait(b & 1)
        return self._row_ptrs[b & 1][r]

    def _wait(self, k: int) -> None:
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None

    def row(self, t: int) -> torch.Tensor:
        b, r = divmod(t, self.block)
        if r == 0:
            self._wait(b & 1)             # the bucket is about to be overwritten
        return self.local[b & 1][r]

    def _launch(self, k: int, rows: Optional[int] = None) -> None:
        rows = self.block if rows is None else rows
        m = rows * self._row_numel
        out = self._flat[k][:self.world * m]
        self._view[k] = out.view((self.world, rows) + tuple(self.local[k].shape[1:]))
        self.pending[k] = self._dist.all_gather_into_tensor(out, self.local[k][:rows].reshape(-1), group=self.group,
                                                            async_op=True)
        self.launched += 1

    def step_done(self, t: int) -> None:
        if (t + 1) % self.block == 0:
            self._launch((t // self.block) & 1)

    def flush(self, t_last: int) -> None:
        """Gather the bucket holding step t_last if it was not closed by step_done: only the rows filled so
        far are sent (a short run, or the tail of a long one, does not pay for a whole bucket)."""
        rows = (t_last + 1) % self.block
        if rows != 0:
            self._launch((t_last // self.block) & 1, rows)

    def result(self, bucket: int) -> torch.Tensor:
        self._wait(bucket & 1)
        return self._view[bucket & 1]

    def warm_up(self) -> None:
        """Run both message shapes once (a whole bucket, a flushed tail) so that the backend's first-use costs -
        communicator channels, buffer registration - are paid before a timed region; leaves the buckets zeroed."""
        for k in range(2):
            self._launch(k)
            self._wait(k)
            self._launch(k, 1)
            self._wait(k)
            self.local[k].zero_()

    def drain(self) -> None:
        self._wait(0)
        self._wait(1)
