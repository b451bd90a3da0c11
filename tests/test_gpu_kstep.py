"""GPU parity of the k-step kernels (fpv_step_n) and the other multi-step entry points (fpv_rollout, the hipGraph replay): bit for bit
what k single-step launches leave, every switch, strided outputs, full size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from gpu_helpers import DEV, _drone_batch, _clone_batch_state
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]


def test_rollout_equals_repeated_step(params_1k):
    n, k = 777, 33
    acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=5)).to(DEV)
    e1, e2 = _drone_batch(params_1k, n), _drone_batch(params_1k, n)
    e1.reset(); e2.reset()
    for t in range(k):
        e1.step(acts[t], return_imu=False)
    e2.rollout(acts)
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.reward, e2.reward)
    # held action: [n,4] + k taken from the outputs
    e1.reset(); e2.reset()
    r = torch.zeros((k, n), dtype=torch.float32, device=DEV)
    for t in range(k):
        e1.step(acts[0], return_imu=False)
    e2.rollout(acts[0].contiguous(), rewards=r)
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(r[-1], e1.reward)


def test_graph_rollout_equals_plain_rollout(params_1k):
    """fpv_rollout_graph: k launches replayed from a cached hipGraph (small, launch-bound batches)."""
    from fpyv_amd.env import DroneBatch
    n, k = 4096, 40
    acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=8)).to(DEV)
    acts2 = (acts * 0.5).contiguous()
    e1, e2 = _drone_batch(params_1k, n), _drone_batch(params_1k, n)
    e1.reset(); e2.reset()
    r1 = torch.zeros((k, n), device=DEV); r2 = torch.zeros((k, n), device=DEV)
    for rep in range(3):                               # same arguments: the cached graph is replayed
        e1.rollout(acts, rewards=r1)
        e2.rollout(acts, rewards=r2, graph=True)
    e1.rollout(acts2); e2.rollout(acts2, graph=True)   # new arguments: the graph is rebuilt
    e1.rollout(acts, rewards=r1); e2.rollout(acts, rewards=r2, graph=True)
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(r1, r2) and torch.equal(e1.done_u8, e2.done_u8)
    assert e1.state_dict()["step_counter"] == e2.state_dict()["step_counter"] == 5 * k
    # stick-noise / fp16 handles are keyed by the per-launch step index, which a graph would freeze: they are served by
    # the k-step kernel, with the same result as k plain launches
    a64 = acts[:, :64].contiguous()
    for kw in (dict(stick_noise=True, noise_seed=3), dict(fp16_state=True)):
        g1, g2 = DroneBatch(params_1k, 64, device=DEV, **kw), DroneBatch(params_1k, 64, device=DEV, **kw)
        g1.reset(); g2.reset()
        for rep in range(2):
            g1.rollout(a64, fused=False)
            g2.rollout(a64, graph=True)
        torch.cuda.synchronize()
        assert torch.equal(g1.state, g2.state) and torch.equal(g1.done_u8, g2.done_u8)
        if g1.state_h is not None:
            assert torch.equal(g1.state_h.view(torch.int16), g2.state_h.view(torch.int16))


def test_step_n_fuzz_bitwise_equal_to_single_steps(params_1k):
    """The fused k-step kernel against k launches of the single-step kernel, over auto-reset, in-kernel
    noise (with and without a base action), ground flag | object list, Kahan rows, held vs per-step
    actions, per-step vs last-step outputs, per-step done-bit rows, episode bookkeeping and ragged n:
    every buffer must come out bit for bit the same."""
    from fpyv_amd.env import DroneBatch
    rng = np.random.default_rng(77)
    objs = ((2, 0.3, -0.2, 0.9, 0.35, 0.0), (1, 1.2, 0.4, 0.0, 0.5, 1.1), (0, 0, 0, 0, 0, 0))
    base = params_1k.replace(init_position=np.array([0.0, 0.0, 0.55]), ceiling=1.6, noise_gain=0.7)
    for case in range(32):
        auto, kahan, noise, track = bool(case & 1), bool(case & 2), bool(case & 4), bool(case & 8)
        world = ("none", "flag", "list")[case % 3]
        held = (case % 5) == 0
        per_step_out = (case % 4) != 3
        n = int(rng.integers(1, 900))
        k = int(rng.integers(1, 48))
        p = base.replace(ground=(world == "flag"))
        kw = dict(auto_reset=auto, kahan_position=kahan, stick_noise=noise, noise_seed=case, with_done_bits=True,
                  track_episodes=track, with_action_out=noise, drone_id_offset=1000 * case)
        a, b = _drone_batch(p, n, **kw), _drone_batch(p, n, **kw)
        pos = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(0.3, 1.2, (n, 1))], axis=1).astype(np.float32)
        a.reset(position=pos, velocity=[0.5, 0, 0])
        _clone_batch_state(b, a)
        acts = rng.uniform(-1, 1, (1 if held else k, n, 4)).astype(np.float32)
        acts[..., 3] = rng.uniform(-1, -0.3, acts.shape[:2])
        act_t = torch.from_numpy(acts).to(DEV)
        no_action = noise and (case % 7) == 4
        words = (n + 63) // 64
        ra, rb = torch.zeros((k, n), device=DEV), torch.zeros((k, n), device=DEV)
        da, db = (torch.zeros((k, n), dtype=torch.uint8, device=DEV) for _ in range(2))
        ba, bb = (torch.zeros((k, words), dtype=torch.int64, device=DEV) for _ in range(2))
        wind = (0.3, -0.2, 0.1)
        a.set_objects(objs if world == "list" else ())       # bound for the following rollouts ...
        b.set_objects(() if world == "list" else [(0, 0, 0, 0, 0, 0)])  # ... and replaced by rollout(object_list=...) below
        a.set_done_bits_target(ba, stride_words=words)
        b.set_done_bits_target(bb, stride_words=words)
        arg = None if no_action else (act_t[0].contiguous() if held else act_t)
        out = dict(rewards=ra, dones=da) if per_step_out else {}
        out_b = dict(rewards=rb, dones=db) if per_step_out else {}
        a.rollout(arg, wind=wind, steps=k, fused=False, **out)
        b.rollout(arg, wind=wind, steps=k, fused=True, object_list=objs if world == "list" else (), **out_b)
        torch.cuda.synchronize()
        tag = f"case {case}: auto={auto} kahan={kahan} noise={noise} track={track} world={world} held={held} n={n} k={k}"
        for name in ("state", "reward", "done_u8", "accel", "noise_state", "pos_comp", "action_out", "ep_return",
                     "ep_length", "last_return", "last_length"):
            x, y = getattr(a, name, None), getattr(b, name, None)
            if x is not None:
                assert torch.equal(x, y), f"{tag}: {name}"
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ba, bb), tag
        assert a.state_dict()["step_counter"] == b.state_dict()["step_counter"] == k
        if per_step_out and auto:
            unpacked = ((bb.cpu().numpy().view(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1))
            assert np.array_equal(unpacked.reshape(k, -1)[:, :n].astype(np.uint8), db.cpu().numpy()), tag


@pytest.mark.parametrize("kind", ["fp16", "racer", "racer_written", "racer_cpid"])
def test_step_n_other_modes_bitwise(params_1k, kind):
    """fp16 storage (the state takes its binary16 round trip in registers every step) and the Racer variants."""
    from fpyv_amd.env import DroneBatch, RacerBatch
    rng = np.random.default_rng(5)
    for n, k in ((1, 7), (333, 40), (4096 + 3, 25)):
        if kind == "fp16":
            p = params_1k.replace(ceiling=10.4)
            mk = lambda: DroneBatch(p, n, device=DEV, fp16_state=True, rounding_seed=9, auto_reset=True, with_done_bits=True)   # noqa: E731
            acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=3)).to(DEV)
        else:
            pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0]])
            p = params_1k.replace(mode=1, racer_pid=pid, racer_omega_dt=(kind == "racer"), ceiling=5e-4)
            if kind == "racer_cpid":
                p = p.replace(racer_pid=-pid, racer_pid_variant=1, pid_integral_clip=0.05, pid_min_output=-0.004,
                              pid_max_output=0.006, pid_derivative_transition_rate=0.3)
            mk = lambda: RacerBatch(p, n, device=DEV, auto_reset=True, with_done_bits=True, track_episodes=True)   # noqa: E731
            acts = torch.from_numpy(np.concatenate([rng.uniform(-6, 6, (k, n, 3)), rng.uniform(0, 8, (k, n, 1))], axis=2).astype(np.float32)).to(DEV)
        a, b = mk(), mk()
        a.reset(); b.reset()
        ra, rb = torch.zeros((k, n), device=DEV), torch.zeros((k, n), device=DEV)
        a.rollout(acts, rewards=ra, fused=False)
        b.rollout(acts, rewards=rb, fused=True)
        a.rollout(acts[: k // 2 + 1], fused=False)          # a second call continues the step counter / rounding seeds
        b.rollout(acts[: k // 2 + 1], fused=True)
        torch.cuda.synchronize()
        assert torch.equal(a.state, b.state) and torch.equal(ra, rb), (kind, n, k)
        assert torch.equal(a.done_u8, b.done_u8) and torch.equal(a.done_bits, b.done_bits) and torch.equal(a.reward, b.reward)
        if kind == "fp16":
            assert torch.equal(a.state_h.view(torch.int16), b.state_h.view(torch.int16))
        else:
            assert torch.equal(a.ep_length, b.ep_length) and torch.equal(a.last_return, b.last_return)
            if n > 1 and k >= 25:
                assert bool((a.last_length > 0).any()), "the ceiling must end some episodes"


@pytest.mark.parametrize("kind", ["f32", "fp16", "racer", "racer_written"])
def test_step_n_strided_done_rows_stay_inside_their_row(params_1k, kind):
    """Per-step done-mask rows (done_bits_stride > 0) of the k-step kernels for every state family, at populations whose
    LAST wave of the grid is wholly dead (n % 128 in 1..64: a 128-thread workgroup launches a second wave that owns no
    drone).  The bucket is [k, words + 1] with stride words + 1 and a sentinel in the extra column: a wave that stores a
    mask word it does not own writes exactly there (or, with stride = words, into the next step's row).  Rows must
    equal those of k single-step launches, sentinels untouched - with and without reward/done leaving per step (the
    quiet loop and the RollOut::step path store the mask in different places)."""
    from fpyv_amd.env import DroneBatch, RacerBatch
    rng = np.random.default_rng(11)
    SENT = -0x0123456789ABCDF
    for n, k in ((1, 9), (64, 8), (4096 + 3, 21), (128 * 7 + 33, 12)):
        assert 1 <= n % 128 <= 64
        if kind in ("f32", "fp16"):
            p = params_1k.replace(ceiling=10.0005)         # 0.5 mm above the start height: the ceiling ends episodes within the k steps
            mk = lambda: DroneBatch(p, n, device=DEV, fp16_state=(kind == "fp16"), rounding_seed=5, auto_reset=True, with_accel=False)   # noqa: E731
            acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=3)).to(DEV)
            acts[..., 3] = torch.from_numpy(rng.uniform(0.2, 1, (k, n)).astype(np.float32)).to(DEV)
        else:
            pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0]])
            p = params_1k.replace(mode=1, racer_pid=pid, racer_omega_dt=(kind == "racer"), ceiling=2e-5)
            mk = lambda: RacerBatch(p, n, device=DEV, auto_reset=True)   # noqa: E731
            acts = torch.from_numpy(np.concatenate([rng.uniform(-6, 6, (k, n, 3)), rng.uniform(0, 8, (k, n, 1))], axis=2).astype(np.float32)).to(DEV)
        words = (n + 63) // 64
        for per_step_out in (False, True):
            a, b = mk(), mk()
            a.reset(); b.reset()
            ba = torch.full((k, words + 1), SENT, dtype=torch.int64, device=DEV)
            bb = torch.full((k, words + 1), SENT, dtype=torch.int64, device=DEV)
            a.set_done_bits_target(ba, stride_words=words + 1)
            b.set_done_bits_target(bb, stride_words=words + 1)
            out_a = dict(dones=torch.zeros((k, n), dtype=torch.uint8, device=DEV)) if per_step_out else {}
            out_b = dict(dones=torch.zeros((k, n), dtype=torch.uint8, device=DEV)) if per_step_out else {}
            a.rollout(acts, fused=False, **out_a)
            b.rollout(acts, fused=True, **out_b)
            torch.cuda.synchronize()
            tag = (kind, n, k, per_step_out)
            assert bool((bb[:, words] == SENT).all()) and bool((ba[:, words] == SENT).all()), f"{tag}: a dead wave stored a mask word"
            assert torch.equal(ba, bb), tag
            assert torch.equal(a.state, b.state), tag
            assert n < 64 or int((bb[:, :words] != 0).sum()) > 0, f"{tag}: the ceiling must set some bits"
            if per_step_out:
                unpacked = ((bb[:, :words].cpu().numpy().view(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1))
                assert np.array_equal(unpacked.reshape(k, -1)[:, :n].astype(np.uint8), out_b["dones"].cpu().numpy()), tag
        # the tight bucket of the collective path (stride == words): the stray word of the old code was row t + 1, word 0
        a, b = mk(), mk()
        a.reset(); b.reset()
        ba, bb = (torch.zeros((k + 1, words), dtype=torch.int64, device=DEV) for _ in range(2))
        ba[k], bb[k] = SENT, SENT
        a.set_done_bits_target(ba, stride_words=words); b.set_done_bits_target(bb, stride_words=words)
        a.rollout(acts, fused=False); b.rollout(acts, fused=True)
        torch.cuda.synchronize()
        assert torch.equal(ba, bb) and bool((bb[k] == SENT).all()), (kind, n, k, "tight")


def test_config2_full_size_1000_steps_fused_noise_vs_oracle(params_1k):
    """BASELINE configs[2] at its full size and length: 2^20 drones x 1000 steps of in-kernel EMA-noise
    sticks.  The applied sticks of 4096 sampled drones (block / wave edges + random) are captured from
    `action_out` every step, replayed through the float64 oracle, and must agree to 1e-5; the fused
    k-step kernel must then reproduce the 1000 single launches bit for bit on all 2^20 drones."""
    from fpyv_amd.env import DroneBatch
    n, T = 1 << 20, 1000
    kw = dict(device=DEV, stick_noise=True, noise_seed=4242, with_accel=False, with_action_out=True)
    env = DroneBatch(params_1k, n, **kw)
    env.reset()
    idx = np.unique(np.concatenate([np.arange(0, 192), np.arange(n - 192, n), np.arange(65536 - 64, 65536 + 64),
                                    np.random.default_rng(1).integers(0, n, 3700)]))[:4096]
    assert len(idx) == 4096
    idx_t = torch.from_numpy(idx).to(DEV)
    acts = torch.zeros((T, len(idx), 4), device=DEV)
    for t in range(T):
        env.step(None, return_imu=False)
        acts[t] = env.action_out[idx_t]
    torch.cuda.synchronize()
    assert 0.05 < float(acts.std()) < 0.5                       # the stationary EMA profile (sigma = 0.229 per channel)
    got = env.state[:, idx_t].cpu().numpy()
    ref = oracle.drone_initial_state(len(idx), params_1k.init_position, params_1k.init_velocity, [0, 0, 0])
    oracle.drone_run(params_1k, ref, acts.cpu().numpy().astype(np.float64), threads=0)
    assert_parity(soa_vs_oracle(np.ascontiguousarray(got), ref, len(idx)), REL_TOL, "configs[2] full size, 1000 steps")
    fused = DroneBatch(params_1k, n, **kw)
    fused.reset()
    fused.rollout(None, steps=T)
    torch.cuda.synchronize()
    assert torch.equal(fused.state, env.state) and torch.equal(fused.noise_state, env.noise_state)
    assert torch.equal(fused.action_out, env.action_out) and torch.equal(fused.reward, env.reward)
