"""Multi-GPU host logic on CPU: world_size-2 gloo runs of the sharding + done-mask all-gather used by
bench.py --gpus N (on the GPU box the same code runs over RCCL).  The per-shard done masks come from
the oracle (checker), so no GPU and no product compute call is involved."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fpyv_amd import load_params, sticks
from fpyv_amd.dist import DoneGather, pack_done_bits, shard_range, shard_sizes, unpack_done_bits


def test_shard_ranges_cover_and_are_contiguous():
    for n, w in [(8, 8), (10, 4), (1 << 23, 8), (7, 3), (5, 8)]:
        lo_prev = 0
        for r in range(w):
            lo, hi = shard_range(n, w, r)
            assert lo == lo_prev and hi >= lo
            lo_prev = hi
        assert lo_prev == n and sum(shard_sizes(n, w)) == n
    assert shard_range(1 << 23, 8, 3) == (3 << 20, 4 << 20)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_done_bit_packing_roundtrip():
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 1000, 4096):
        d = torch.from_numpy(rng.integers(0, 2, n).astype(np.uint8))
        bits = pack_done_bits(d)
        assert bits.dtype == torch.int64 and bits.numel() == (n + 63) // 64
        assert torch.equal(unpack_done_bits(bits, n), d)
        ref = np.zeros((n + 63) // 64, dtype=np.uint64)     # same layout as the kernel's wave ballot
        for i in np.flatnonzero(d.numpy()):
            ref[i // 64] |= np.uint64(1) << np.uint64(i % 64)
        assert np.array_equal(bits.numpy().view(np.uint64), ref)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, steps, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        p = load_params(fps=1000).replace(init_position=np.array([0.0, 0.0, 0.2]))
        lo, hi = shard_range(n_total, world, rank)
        n = hi - lo
        # global-id keyed sticks: drone i sees the same stream whatever shard it lands in
        acts = sticks.ema_noise(steps, range(lo, hi), seed=3).astype(np.float64)
        acts[..., 3] -= 0.9                                   # low throttle: many drones reach the ground
        st = oracle.drone_initial_state(n, p.init_position, p.init_velocity, [0, 0, 0])
        block = 7                                              # does not divide `steps`: exercises flush()
        g = DoneGather(((n + 63) // 64,), torch.int64, "cpu", block=block)
        seen = []
        for t in range(steps):
            _, _, done = oracle.drone_run(p, st, acts[t:t + 1])
            g.row(t).copy_(pack_done_bits(torch.from_numpy(done)))
            g.step_done(t)
            if (t + 1) % block == 0:
                seen.append(g.result(t // block).clone())         # [world, block, words]
        g.flush(steps - 1)
        tail = g.result((steps - 1) // block).clone()
        assert tail.shape[1] == steps % block, "a flushed bucket ships only its filled rows"
        seen.append(tail)
        g.drain()
        seen = [torch.cat(seen, dim=1).permute(1, 0, 2)]       # -> [steps, world, words]
        np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), seen[0].contiguous().numpy())
    finally:
        dist.destroy_process_group()


def test_world2_gloo_done_allgather_matches_single_process(tmp_path):
    from oracle import oracle
    world, n_total, steps = 2, 256, 260
    mp.spawn(_worker, args=(world, _free_port(), n_total, steps, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(tmp_path / "gathered_0.npy")
    g1 = np.load(tmp_path / "gathered_1.npy")
    assert np.array_equal(g0, g1), "every rank must see the same global mask"
    # single-process reference over all drones
    p = load_params(fps=1000).replace(init_position=np.array([0.0, 0.0, 0.2]))
    acts = sticks.ema_noise(steps, range(n_total), seed=3).astype(np.float64)
    acts[..., 3] -= 0.9
    st = oracle.drone_initial_state(n_total, p.init_position, p.init_velocity, [0, 0, 0])
    any_done = False
    for t in range(steps):
        _, _, done = oracle.drone_run(p, st, acts[t:t + 1])
        any_done |= bool(done.any())
        got = np.concatenate([unpack_done_bits(torch.from_numpy(g0[t, r]), n_total // world).numpy()
                              for r in range(world)])
        assert np.array_equal(got, done), f"step {t}: gathered mask != concatenation of shard masks"
    assert any_done, "the scenario must actually produce done flags"


def _choose_worker(rank, world, port, stub, out_dir, ipc):
    import json
    from fpyv_amd.dist import IPC_ENV, choose_backend
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if ipc is None:
        os.environ.pop(IPC_ENV, None)
    else:
        os.environ[IPC_ENV] = ipc
    c = choose_backend(limit_s=20.0, stub=stub)
    # the default group is the gloo control plane the ranks agreed on: it must still work for the caller
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    c["sum"] = int(t)
    with open(os.path.join(out_dir, f"choice_{rank}.json"), "w") as f:
        json.dump(c, f)
    dist.destroy_process_group()


@pytest.mark.parametrize("stub,ipc", [("ok", "1"), ("fail", None)])
def test_choose_backend_world2(tmp_path, stub, ipc):
    """A program that is its own rank process (examples/sharded_vec_env.py) decides before it touches the GPU whether
    RCCL comes up: preflight in fresh children (here the gloo stub), agreement over gloo, the caller's IPC mode first."""
    import json
    mp.spawn(_choose_worker, args=(2, _free_port(), stub, str(tmp_path), ipc), nprocs=2, join=True)
    c0, c1 = (json.load(open(tmp_path / f"choice_{r}.json")) for r in range(2))
    assert c0["backend"] == c1["backend"] and c0["sum"] == c1["sum"] == 3
    if stub == "ok":
        assert c0["backend"] == "nccl" and c0["ipc_mode"] == "1" and c0["fallback_reason"] is None and len(c0["preflight"]) == 1
    else:
        assert c0["backend"] == "gloo" and "preflight failed" in c0["fallback_reason"]
        assert [p["ipc_mode"] for p in c0["preflight"]] == ["0", "1"]          # the default first, then the other value
