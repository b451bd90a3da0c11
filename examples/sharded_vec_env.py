#!/usr/bin/env python3
"""A population sharded over the GPUs of one node: one process per GPU, contiguous shards, no collective in the physics,
one bucketed all-gather of the bit-packed done mask for whoever needs the global view (BASELINE configs[4]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 \\
        examples/sharded_vec_env.py --drones 8388608 --steps 512

Every drone's stick noise is keyed by its GLOBAL id (`drone_id_offset` = the shard's first drone), so the run is the
same whichever way the population is cut - `--check` replays rank 0's shard alone afterwards and compares bit for bit.
`--all-ranks-on-gpu0 --backend gloo` rehearses the same program on a one-GPU machine (at most 6 ranks).

`--backend auto` (default) does not bet the job on RCCL coming up: before this process touches the GPU, every rank runs
an RCCL preflight in a fresh child process (fpyv_amd.dist.choose_backend: init + one all-gather, wall limit; first with
the caller's HSA_ENABLE_IPC_MODE_LEGACY, then with the other value) and the ranks agree over gloo; if RCCL does not
start, the masks travel over gloo (host-staged) and the program says so.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.dist import IPC_ENV, DoneGather, choose_backend, shard_range, unpack_done_bits  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drones", type=int, default=1 << 22, help="the WHOLE population")
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--block", type=int, default=64, help="steps per all-gather bucket")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help='"nccl" = RCCL over xGMI; "gloo" for a rehearsal; "auto" = RCCL if a preflight brings it up, else gloo')
    ap.add_argument("--preflight-timeout-s", type=float, default=60.0)
    ap.add_argument("--all-ranks-on-gpu0", action="store_true")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    os.environ.setdefault(IPC_ENV, "0")       # this pool's driver only has dmabuf IPC; a caller's value wins
    rank, world = int(os.environ.setdefault("RANK", "0")), int(os.environ.setdefault("WORLD_SIZE", "1"))
    local = 0 if a.all_ranks_on_gpu0 else int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    backend, group = a.backend, None
    if backend == "auto" and (world == 1 or a.all_ranks_on_gpu0):
        backend = "gloo" if a.all_ranks_on_gpu0 else "nccl"       # RCCL refuses two ranks on one device; one rank needs no preflight
    if backend == "auto":
        # nothing has touched the GPU yet: the preflight children do, this process only talks gloo so far
        choice = choose_backend(limit_s=a.preflight_timeout_s, log=lambda m: print(m, file=sys.stderr))
        backend = choice["backend"]
        if choice["ipc_mode"] is not None:
            os.environ[IPC_ENV] = choice["ipc_mode"]
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        # the default group stays gloo (the control plane the ranks agreed on); the masks get their own RCCL group
        group = dist.new_group(backend="nccl") if backend == "nccl" else None
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))          # not the null stream: see INTEGRATION.md 3

    lo, hi = shard_range(a.drones, world, rank)
    n = hi - lo
    assert all(shard_range(a.drones, world, r)[1] - shard_range(a.drones, world, r)[0] == n for r in range(world)), \
        "DoneGather wants equal shards: pick a population that divides by the number of ranks"
    params = load_params(fps=1000, ceiling=10.05, noise_gain=3.0)   # tight ceiling + strong sticks: episodes end often
    kw = dict(device=dev, auto_reset=True, stick_noise=True, noise_seed=7, with_accel=False, with_done_bits=True)
    env = DroneBatch(params, n, drone_id_offset=lo, **kw)
    env.reset()
    words = (n + 63) // 64
    gather = DoneGather((words,), torch.int64, dev, block=a.block, group=group)
    gather.warm_up()

    ended = torch.zeros((), dtype=torch.int64, device=dev)         # episodes ended anywhere in the job, from the gathered masks
    t = 0
    while t < a.steps:
        span = min(a.block - t % a.block, a.steps - t)
        env.set_done_bits_target(gather.row_ptr(t), stride_words=words)      # step t's mask -> row t % block of the bucket
        env.rollout(None, steps=span)                                         # ONE launch for the span (k-step kernel)
        t += span
        if t % a.block == 0:
            gather.step_done(t - 1)                                           # the bucket travels while the next span runs
            if t >= 2 * a.block:                                              # consume the bucket before the one in flight
                ended += gather.result(t // a.block - 2).ne(0).sum()          # (any consumer: here just "words with a done bit")
    gather.flush(a.steps - 1)
    gather.drain()
    last = gather.result((a.steps - 1) // a.block)                            # [world, rows, words] of the last bucket
    row = (a.steps - 1) % a.block
    done_now = sum(int(unpack_done_bits(last[r, row], n).sum()) for r in range(world))
    torch.cuda.synchronize()
    mine = int(env.done.sum())
    tot = torch.tensor([mine], device=dev)
    dist.all_reduce(tot, group=group)
    if rank == 0:
        print(f"{world} ranks x {n} drones over {backend}, {a.steps} steps: done in the last step, from the gathered masks {done_now}, "
              f"from the ranks' own flags {int(tot)} (must agree); mask words with an ended episode in the consumed buckets {int(ended)}; "
              f"collectives launched {gather.launched}")
        assert done_now == int(tot)
    if a.check and rank == 0:
        whole = DroneBatch(params, a.drones, **kw)                            # the same drones as ONE batch
        whole.reset()
        whole.rollout(None, steps=a.steps)
        torch.cuda.synchronize()
        assert torch.equal(whole.state[:, lo:hi], env.state[:, :n]), "a shard must reproduce its slice of the whole batch"
        print("rank 0's shard equals its slice of the unsharded run bit for bit")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
