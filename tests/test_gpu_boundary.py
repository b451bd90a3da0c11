"""The drop-in boundary on the GPU: the reference's call sequence through fpyv_amd.components, the gym-style env, split phase, argument
errors, checkpoints, the INTEGRATION.md stub, a plain-C host, host threads, device binding, the multi-rank rehearsals on one GPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from gpu_helpers import DEV, _drone_batch, _two_host_threads_two_handles
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]


def test_broadcast_action_and_numpy_action(params_1k):
    env = _drone_batch(params_1k, 100)
    env.reset()
    env.step(np.array([0.5, 0, 0, 0]), return_imu=False)          # simulator.py:89 style single action
    env2 = _drone_batch(params_1k, 100)
    env2.reset()
    env2.step(torch.tensor([[0.5, 0, 0, 0]] * 100, device=DEV), return_imu=False)
    assert torch.equal(env.state, env2.state)


def test_vec_env_surface(params_1k):
    from fpyv_amd.env import FpvVecEnv
    env = FpvVecEnv(params_1k.replace(ceiling=50.0), num_envs=1024, device=DEV)
    obs = env.reset()
    assert obs.shape == (1024, 13) and obs.data_ptr() == env.batch.state.data_ptr()   # zero-copy view
    a = torch.zeros((1024, 4), device=DEV)
    obs, reward, done, info = env.step(a)
    assert obs.shape == (1024, 13) and reward.shape == (1024,) and done.shape == (1024,)
    # done[N] bool - the tensor the kernel itself writes (one byte of 0/1 per drone), not a converted copy
    assert done.dtype == torch.bool and done.data_ptr() == env.batch.done_u8.data_ptr()
    assert "episode_return" in info and info["episode_length"].dtype == torch.int32
    torch.cuda.synchronize()
    np.testing.assert_allclose(reward.cpu().numpy(), -np.linalg.norm(obs[:, 0:3].cpu().numpy() - params_1k.goal, axis=1), rtol=1e-5, atol=1e-6)
    # done really is written as a bool: drive every other drone through the ceiling
    low = FpvVecEnv(params_1k.replace(ceiling=10.0005), num_envs=130, device=DEV, auto_reset=True)
    low.reset()
    a = torch.zeros((130, 4), device=DEV)
    a[::2, 3] = 1.0                                    # full throttle: climbs past 10.0005 m within a few steps
    a[1::2, 3] = -0.9                                  # 5 % throttle: sinks
    hits = torch.zeros(130, dtype=torch.bool, device=DEV)
    for _ in range(40):
        _, _, done, _ = low.step(a)
        assert done.dtype == torch.bool
        hits |= done                                   # bool arithmetic on the kernel's own output
    assert bool(hits[::2].all()) and not bool(hits[1::2].any())
    raw = low.batch.done_u8.cpu().numpy()
    assert set(np.unique(raw)) <= {0, 1}


@pytest.mark.parametrize("case", ["plain2", "noise3", "racer2", "objects2", "fp16_2"])
def test_split_phase_partitions_are_bitwise_the_single_batch(params_1k, case):
    """FpvVecEnv(partitions=P): step_async(part, action) / step_wait(part) - each partition its own handle, stream and
    kernel chain over column ranges of the SAME tensors, drones keyed by their global id.  Closed loop (a linear policy
    on each partition's observation view, computed on the caller's stream while the other partition steps) for 200
    steps with in-kernel auto-reset: every buffer equals the unpartitioned env driven by the same policy, bit for bit -
    state, reward, done, the bit-packed mask, episode bookkeeping, the noise rows, the applied sticks."""
    from fpyv_amd.env import FpvVecEnv, partition_bounds
    from fpyv_amd.objects import Cylinder, Ground
    n, T = 128 * 37 + 55, 200                       # the last partition ends in a ragged wave
    parts = 3 if case == "noise3" else 2
    p = params_1k.replace(ceiling=10.3, noise_gain=0.8)
    kw = dict(num_envs=n, device=DEV, auto_reset=True, track_episodes=True, with_done_bits=True)
    if case == "noise3":
        kw.update(stick_noise=True, noise_seed=99, drone_id_offset=5000, with_action_out=True)
    if case == "racer2":
        pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0]])
        p = params_1k.replace(mode=1, racer_pid=pid, ceiling=3e-3)
        kw.update(mode="racer")
    if case == "fp16_2":
        kw.update(fp16_state=True, rounding_seed=21, drone_id_offset=777)      # the rounding stream is keyed by the global id too
    if case == "objects2":
        p = p.replace(init_position=np.array([0.0, 0.0, 0.12]), init_velocity=np.array([1.0, 0.0, -3.0]), ceiling=3.0)   # diving: the ground ends episodes
        kw.update(object_list=[Ground(), Cylinder(position=[1.5, 0.2, 0.0], radius=0.4, height=1.0)], wind=(0.4, -0.1, 0.0))
    one, split = FpvVecEnv(p, **kw), FpvVecEnv(p, partitions=parts, **kw)
    assert split.partitions == parts and [split.partition_range(k) for k in range(parts)] == list(partition_bounds(n, parts))
    assert all(lo % 128 == 0 for lo, _ in partition_bounds(n, parts)) and partition_bounds(n, parts)[-1][1] == n
    torch.manual_seed(3)
    W = torch.randn(4, 13, device=DEV) * (0.02 if case != "racer2" else 0.5)
    bias = torch.tensor([0.0, 0.0, 0.0, 3.0 if case == "racer2" else -0.9 if case == "objects2" else 0.4], device=DEV)
    policy = lambda o: torch.tanh(o @ W.t()) + bias        # noqa: E731  ([n, 13] view -> [n, 4] rows)
    obs = one.reset()
    split.reset()
    for _ in range(T):
        obs, _, _, _ = one.step(policy(obs).contiguous())
    hits = 0
    for t in range(T):
        for k in range(parts):
            o, r, d, info = split.step_wait(k)                 # views of this partition's columns after ITS last step
            split.step_async(k, policy(o).contiguous())        # the other partition's step is in flight meanwhile
    for k in range(parts):
        o, r, d, info = split.step_wait(k)
        lo, hi = split.partition_range(k)
        assert o.shape == (hi - lo, 13) and r.shape == (hi - lo,) and d.dtype == torch.bool and info["episode_length"].shape == (hi - lo,)
        if case == "fp16_2":      # a decoded copy of the partition's columns: equal to the same columns of the whole batch's decoding
            torch.cuda.synchronize()
            assert torch.equal(o, split.batch.rows_f32(0, 13)[lo:hi])
        else:
            assert o.data_ptr() == split.batch.state.data_ptr() + 4 * lo          # a view, not a copy
    torch.cuda.synchronize()
    a, b = one.batch, split.batch
    for name in ("state", "state_h", "reward", "done_u8", "done_bits", "ep_return", "ep_length", "last_return", "last_length", "noise_state", "action_out"):
        x, y = getattr(a, name, None), getattr(b, name, None)
        if x is not None:
            assert torch.equal(x.view(torch.int16) if name == "state_h" else x, y.view(torch.int16) if name == "state_h" else y), (case, name)
    assert int(a.last_length.max()) > 0, "auto-reset must have ended episodes"
    # step(): all partitions at once, still the same bits; a checkpoint of the split env continues in an unpartitioned one
    act = (torch.rand((n, 4), device=DEV) * 2 - 1) * (1.0 if case != "racer2" else 4.0)
    one.step(act); split.step(act)
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state) and torch.equal(a.done_u8, b.done_u8)
    if case == "fp16_2":
        assert torch.equal(a.state_h.view(torch.int16), b.state_h.view(torch.int16))
    # the mask redirected to a caller's row (what a collective's bucket is): every partition writes its own words of it
    row_a, row_b = (torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device=DEV) for _ in range(2))
    one.batch.set_done_bits_target(row_a); split.set_done_bits_target(row_b)
    one.step(act); split.step(act)
    torch.cuda.synchronize()
    assert torch.equal(row_a, row_b) and torch.equal(a.state, b.state)
    one.batch.set_done_bits_target(None); split.set_done_bits_target(None)
    if case == "noise3":
        # a reset in the middle of a run: the stick-noise streams are keyed by the step counter, which runs on across a reset
        # in the single batch - and must in every partition
        one.reset(); split.reset()
        for _ in range(3):
            one.step(act); split.step(act)
        torch.cuda.synchronize()
        assert torch.equal(a.state, b.state) and torch.equal(a.noise_state, b.noise_state) and torch.equal(a.action_out, b.action_out)
    ck = split.state_dict()
    steps_done = T + 2 + (3 if case == "noise3" else 0)
    assert ck["partition_step_counters"] == [steps_done] * parts and ck["step_counter"] == steps_done
    third = FpvVecEnv(p, **kw)
    third.reset()
    third.batch.load_state_dict({k: v for k, v in ck.items() if k != "partition_step_counters"})
    one.step(act); third.step(act)
    torch.cuda.synchronize()
    assert torch.equal(a.state, third.batch.state)
    for e in (one, split, third):
        e.close()


def test_split_phase_whole_population_calls_are_ordered_after_steps_in_flight(params_1k):
    """`step_async(k, a)` on every partition and then - with NO step_wait - `reset(mask)` / `load_state_dict` / `state_dict`:
    the whole-population call is ordered after the partitions' chains on the device (and the next step_async after it), so
    the result is bit for bit the single batch doing step-then-reset (gym raises here, this API orders).
    2^20 drones and eight queued steps per partition: ~100 us of kernels are still in flight when the reset is enqueued.
    Env convention: /root/reference/tests/rotation_pid.py:57-78."""
    from fpyv_amd.env import FpvVecEnv
    n, parts, depth = 1 << 20, 2, 8
    p = params_1k.replace(ceiling=10.3)
    kw = dict(num_envs=n, device=DEV, auto_reset=True, track_episodes=True, with_done_bits=True)
    one, split = FpvVecEnv(p, **kw), FpvVecEnv(p, partitions=parts, **kw)
    g = torch.Generator(device=DEV); g.manual_seed(12)
    acts = (torch.rand((depth, n, 4), device=DEV, generator=g) * 2 - 1)
    mask = torch.rand(n, device=DEV, generator=g) < 0.37
    one.reset(); split.reset()
    torch.cuda.synchronize()

    def equal(tag):
        torch.cuda.synchronize()
        for name in ("state", "reward", "done_u8", "done_bits", "ep_return", "ep_length", "last_return", "last_length"):
            assert torch.equal(getattr(one.batch, name), getattr(split.batch, name)), (tag, name)

    def burst():
        for t in range(depth):
            one.step(acts[t])
        for t in range(depth):
            for k in range(parts):
                lo, hi = split.partition_range(k)
                split.step_async(k, acts[t, lo:hi], ready=True)

    burst()
    one.reset(mask); split.reset(mask)                 # no step_wait: the chains are still running
    equal("reset(mask) right after step_async")
    moved = (one.batch.state[:3, :n].t() != torch.tensor(p.init_position, device=DEV, dtype=torch.float32)).any(dim=1)
    assert bool((~moved[mask]).all()) and bool(moved[~mask].all()), "masked drones sit at the initial position, the others flew on"
    burst()                                            # and the partitions' next steps come after the reset
    equal("steps after the reset")
    ck = one.state_dict()                              # a checkpoint of the single env at this point
    ck_split = split.state_dict()                      # state_dict right after step_async: ordered after the chains too
    torch.cuda.synchronize()
    assert torch.equal(ck["state"], ck_split["state"]) and ck_split["partition_step_counters"] == [2 * depth] * parts
    burst()
    one.load_state_dict(ck); split.load_state_dict(ck_split)      # no step_wait before the load either
    equal("load_state_dict right after step_async")
    burst()
    equal("steps after the load")
    # host-side whole-population setters reach every partition: wind is read on every step, set_params updates every handle
    one.wind = split.wind = (1.5, -0.5, 0.25)
    p2 = p.replace(mass=p.mass * 1.1)
    one.batch.set_params(p2); split.set_params(p2)
    burst()
    equal("wind and set_params")
    # what the single batch accepts as sticks, step() of the split env accepts too: a list broadcast, a NumPy array, float64
    for a in ([0.1, -0.2, 0.3, 0.4], acts[0].cpu().numpy(), acts[1].double(), acts[2][:, [1, 0, 2, 3]].t().contiguous().t()):
        one.step(a); split.step(a)
    equal("coerced actions")
    split.step_async(0, acts[0, :split.partition_range(0)[1]], ready=True)
    split.close()                                      # a close with a step in flight drains the chain first
    one.close()


def test_split_phase_api_errors(params_1k):
    from fpyv_amd.env import FpvVecEnv
    env = FpvVecEnv(params_1k, num_envs=1000, device=DEV)
    with pytest.raises(RuntimeError):
        env.step_async(0, torch.zeros((1000, 4), device=DEV))
    two = FpvVecEnv(params_1k, num_envs=1000, device=DEV, partitions=2)
    two.reset()
    lo, hi = two.partition_range(1)
    with pytest.raises(ValueError):
        two.step_async(1, torch.zeros((1000, 4), device=DEV))               # the partition's own slice is what it takes
    full = torch.zeros((4, 1000), device=DEV)
    two.step_async(1, full[:, lo:hi])                                        # SoA column slice of a full-size tensor
    two.step_async(0, torch.zeros((1000, 4), device=DEV)[:lo], ready=True)
    assert FpvVecEnv(params_1k, num_envs=100, device=DEV, partitions=4).partitions == 1   # too small to cut: one workgroup
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["g7_racer_main", "g8_racer_pid_thrust"])
def test_vec_env_racer_mode_vs_reference_capture(params_1k, name):
    """FpvVecEnv(mode="racer"): the gym surface over Racer.step (racer_drone_test.py:95-103) against the reference
    captures G7 / G8, stepping through env.step() one call per step; obs = (p, v, q, omega) zero-copy, done is bool."""
    from fpyv_amd.env import FpvVecEnv
    g = load_golden(name)
    p = params_1k.replace(mode=1, racer_pid=g["pid"])
    env = FpvVecEnv(p, num_envs=3, device=DEV, mode="racer", auto_reset=False)
    obs = env.reset()
    assert obs.shape == (3, 13) and obs.data_ptr() == env.batch.state.data_ptr()
    acts = torch.from_numpy(g["actions"]).to(DEV)            # [T, 1, 4]
    snaps = {int(t): k for k, t in enumerate(np.asarray(g["snap_steps"]).reshape(-1))}
    worst = dict(quat=0.0, pos=0.0)
    for t in range(acts.shape[0]):
        obs, reward, done, info = env.step(acts[t].expand(3, 4).contiguous())
        if t + 1 in snaps:
            k = snaps[t + 1]
            o = obs.cpu().numpy().astype(np.float64)
            x, y, z, w = g["quat_xyzw"][0, k]
            qr = np.array([w, x, y, z])
            q = o[0, 6:10] * np.sign(o[0, 6:10] @ qr)
            pr = g["position"][0, k]
            worst["quat"] = max(worst["quat"], np.abs(q - qr).max())
            worst["pos"] = max(worst["pos"], np.abs(o[0, 0:3] - pr).max() / max(np.abs(pr).max(), 1e-3))
            assert np.array_equal(o[0], o[1]) and np.array_equal(o[0], o[2])
    assert done.dtype == torch.bool and not bool(done.any()) and reward.shape == (3,)
    assert worst["quat"] < REL_TOL and worst["pos"] < REL_TOL, worst
    # the same steps through RacerBatch.rollout (k-step kernel) land on the same bits
    from fpyv_amd.env import RacerBatch
    rb = RacerBatch(p, 3, device=DEV)
    rb.reset()
    rb.rollout(acts.expand(-1, 3, 4).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(rb.state, env.batch.state)


def test_argument_errors(params_1k):
    import ctypes as C
    env = _drone_batch(params_1k, 64)
    with pytest.raises(ValueError):
        env.step(torch.zeros((63, 4), device=DEV))
    with pytest.raises((TypeError, ValueError)):
        env.step(torch.zeros((64, 4), device=DEV), object_list=[object()])
    with pytest.raises(ValueError):
        env.step(None)
    L = _lib.lib()
    b = _lib.FpvBuffers()
    C.memmove(C.byref(b), C.byref(env._buf), C.sizeof(b))
    b.action = torch.zeros((64, 4), device=DEV).data_ptr()
    b.ld = 63
    rc = L.fpv_step(env._handle, C.byref(b), None)
    assert rc == -4 and b"ld" in L.fpv_last_error()
    b.ld = env.ld
    b.state = env.state.data_ptr() + 4
    assert L.fpv_step(env._handle, C.byref(b), None) == -4
    b.state = None
    assert L.fpv_step(env._handle, C.byref(b), None) == -1
    bad = _lib.pack_params(params_1k.replace(dt=0.0))
    h = C.c_void_p()
    assert L.fpv_create(C.byref(bad), 8, 0, C.byref(h)) == -5
    assert L.fpv_create(C.byref(_lib.pack_params(params_1k)), 8, 99, C.byref(h)) == -3
    assert not hasattr(L, "fpv_set_tuning"), "removed in ABI 4 (the rejected launch geometries are no longer built)"


def test_checkpoint_resume_is_bit_exact(params_1k, tmp_path):
    """state_dict()/load_state_dict(): tensors + step counter; a resumed run (in-kernel stick noise,
    auto-reset, episode bookkeeping) continues bit for bit, also through torch.save/torch.load."""
    from fpyv_amd.env import DroneBatch
    kw = dict(device=DEV, stick_noise=True, noise_seed=11, auto_reset=True, track_episodes=True, with_accel=False)
    p = params_1k.replace(ceiling=10.3, noise_gain=2.0)
    a, b = DroneBatch(p, 5000, **kw), DroneBatch(p, 5000, **kw)
    a.reset(); b.reset()
    a.rollout(None, steps=120)
    ck = a.state_dict()
    torch.save(ck, tmp_path / "ckpt.pt")
    a.rollout(None, steps=80)
    b.load_state_dict(torch.load(tmp_path / "ckpt.pt", weights_only=True))
    b.rollout(None, steps=80)
    torch.cuda.synchronize()
    for k in ("state", "noise_state", "ep_return", "ep_length", "last_return", "reward", "done_u8"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.state_dict()["step_counter"] == b.state_dict()["step_counter"] == 200
    with pytest.raises(ValueError):
        DroneBatch(p, 4999, **kw).load_state_dict(ck)
    # a checkpoint says what its bits mean: an fp16 state written with another storage encoding (ABI <= 4 recorded none) is
    # refused instead of decoded as garbage; another stick-noise generator is a warning (the run goes on, not bit for bit)
    from fpyv_amd import _lib as _l
    assert ck["abi_version"] == _l.FPV_ABI_VERSION and "philox4x32-7" in ck["noise_generator"]
    with pytest.warns(RuntimeWarning, match="stick-noise generator"):
        b.load_state_dict({k: v for k, v in ck.items() if k != "noise_generator"})
    h = DroneBatch(p, 640, device=DEV, fp16_state=True, with_accel=False)
    h.reset()
    h.rollout(torch.zeros((3, 640, 4), device=DEV))
    ckh = h.state_dict()
    h.load_state_dict(ckh)
    with pytest.raises(ValueError, match="storage encoding"):
        h.load_state_dict({k: v for k, v in ckh.items() if k != "state_h_encoding"})


def test_vec_env_options_pass_through(params_1k):
    from fpyv_amd.env import FpvVecEnv
    from fpyv_amd.objects import Ground
    low = params_1k.replace(init_position=np.array([0.0, 0.0, 0.3]))
    # in-kernel noise sticks over the ground-plane flag (FPV_FLAG_GROUND lives in the common lane function)
    env = FpvVecEnv(low.replace(ground=True), num_envs=256, device=DEV, auto_reset=False, stick_noise=True,
                    noise_seed=3, with_action_out=True)
    env.reset()
    for _ in range(300):
        obs, reward, done, info = env.step(None)
    torch.cuda.synchronize()
    assert obs.shape == (256, 13) and bool(torch.isfinite(obs).all())
    assert 0 < float(env.batch.action_out.abs().max()) <= 1.0
    # a collision world given as object_list: same physics as the ground flag for [Ground()]
    e1 = FpvVecEnv(low, num_envs=64, device=DEV, object_list=[Ground()], auto_reset=False)
    e2 = FpvVecEnv(low.replace(ground=True), num_envs=64, device=DEV, auto_reset=False)
    e1.reset(); e2.reset()
    a = torch.zeros((64, 4), device=DEV); a[:, 3] = -0.8
    for _ in range(400):
        e1.step(a); e2.step(a)
    torch.cuda.synchronize()
    assert torch.equal(e1.batch.state, e2.batch.state)
    # features are orthogonal: object_list x in-kernel noise x Kahan rows == ground flag x noise x Kahan rows
    kw = dict(num_envs=64, device=DEV, auto_reset=True, stick_noise=True, noise_seed=5, kahan_position=True)
    e3 = FpvVecEnv(low, object_list=[Ground()], **kw)
    e4 = FpvVecEnv(low.replace(ground=True), **kw)
    e3.reset(); e4.reset()
    for _ in range(300):
        e3.step(None); e4.step(None)
    torch.cuda.synchronize()
    assert torch.equal(e3.batch.state, e4.batch.state) and torch.equal(e3.batch.pos_comp, e4.batch.pos_comp)
    with pytest.raises(_lib.FpvError):                  # documented restriction: a Ground entry replaces the flag
        FpvVecEnv(low.replace(ground=True), num_envs=8, device=DEV, object_list=[Ground()]).step(a[:8])


def test_set_params_on_a_live_handle(params_1k):
    """fpv_set_params: swap the drone type mid-run (domain randomisation); equals a fresh handle with the
    new parameters started from the same state; layout-changing switches are refused."""
    n = 500
    acts = torch.from_numpy(sticks.ema_noise(60, range(n), seed=4)).to(DEV)
    heavy = params_1k.replace(mass=1.1, max_rates=350.0, drag_coefficients=np.array([2.2, 2.0, 1.0]))
    a = _drone_batch(params_1k, n)
    a.reset()
    a.rollout(acts[:30])
    mid = a.state.clone()
    a.set_params(heavy)
    a.rollout(acts[30:])
    b = _drone_batch(heavy, n)
    b.state.copy_(mid)
    b.rollout(acts[30:])
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state)
    assert not torch.equal(a.state, mid)
    with pytest.raises(_lib.FpvError):
        a.set_params(heavy.replace(mode=1))
    with pytest.raises(_lib.FpvError, match="dt"):
        a.set_params(heavy.replace(dt=-1.0))


def test_soa_action_layout_equals_row_layout(params_1k):
    """Sticks given as [4, n] (the layout of `W[4,13] @ obs[13,n]`) are consumed in place and give
    exactly the step of the [n, 4] layout; a closed policy loop therefore needs no transpose kernels."""
    n, k = 3001, 25
    rows = torch.from_numpy(sticks.ema_noise(k, range(n), seed=6)).to(DEV)       # [k, n, 4]
    e1, e2 = _drone_batch(params_1k, n), _drone_batch(params_1k, n)
    e1.reset(); e2.reset()
    ld = e2.ld
    soa = torch.zeros((4, ld), device=DEV)
    for t in range(k):
        e1.step(rows[t], return_imu=False)
        soa[:, :n] = rows[t].t()
        e2.step(soa[:, :n], return_imu=False)          # a [4, n] view with row stride ld
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.reward, e2.reward)
    # closed loop: a linear policy on the zero-copy SoA observation
    torch.manual_seed(11)
    W = torch.randn(4, 13, device=DEV) * 0.02
    for t in range(50):
        obs_soa = e2.state[:13, :n]                    # [13, n], no copy
        e2.step(torch.tanh(W @ obs_soa), return_imu=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(e2.state).all())


def test_simulator_call_sequence_through_components(params_1k):
    """Drop-in check: the reference's own call sequence - src/core/simulator.py:53-59 (construction from the
    params dict, world objects with their reference constructor arguments, reset), :85-91 (object_list with
    gates, target update, action) and :156 (drone.step(action, wind_velocity_vector, object_list)) - executed
    against fpyv_amd.components with num_envs=1 must land on the reference captures G10 (objects, moving
    target) and G2 (free flight)."""
    import yaml
    from fpyv_amd.components import Cylinder, Drone, Gate, Ground, Target
    from fpyv_amd.params import DEFAULT_PARAMS_PATH
    with open(DEFAULT_PARAMS_PATH) as f:
        params = yaml.safe_load(f)                                    # the params.yaml-shaped dict of simulator.py:9
    params["simulator"]["fps"] = 1000
    params["camera"] = {"camera_angle": 35.0}                         # sections the stepper does not use are ignored
    frozen = yaml.safe_dump(params)

    g = load_golden("g10_objects")
    T, n = g["actions"].shape[:2]
    first = lambda d: int(np.argmax(d)) if d.any() else -1      # noqa: E731
    for k in range(n):
        drone = Drone(params, num_envs=1, device=DEV)                                              # simulator.py:53
        targets = [Target(np.array([0.0, -6.0, 3.0]), 0.8, 1, {"radius": 1.5, "resolution": 20000})]   # :54, generators.py:22-25
        obstacles = [Cylinder(np.array([3.0, 0.0, 0.0]), 1.0, 5.0, 4, 2, random=False),           # :56, generators.py:33-37
                     Cylinder(np.array([-2.0, 2.5, 0.0]), 0.6, 1.5, 4, 2, random=False)]
        gates = [Gate(np.array([4.0, 0.0, 2.5]), np.eye(3), 2.5, shape="circle", resolution=17)]   # :57
        ground = Ground(size=60, resolution=4, random=False)                                       # :58
        drone.reset(position=g["init_position"][k], velocity=g["init_velocity"][k], ypr=g["init_ypr"][k])   # :59
        wind_velocity_vector = np.array([0, 0, 0])                                                 # :63
        dones = []
        for i in range(T):                                                                         # :83
            object_list = [*targets, *gates, *obstacles, ground]                                   # :85
            [target.update() for target in targets]                                                # :87
            action = g["actions"][i, k]                                                            # :89
            ret = drone.step(action=action, wind_velocity_vector=wind_velocity_vector, object_list=object_list)   # :156
            dones.append(drone.done_u8.clone())
        seq = torch.stack(dones).cpu().numpy()[:, 0]
        assert first(seq) == first(g["done"][k]), (k, first(seq), first(g["done"][k]))         # crash on the reference's step
        if not g["done"][k].any():
            ref = np.concatenate([g["state"][k:k + 1, -1], g["R"][k:k + 1, -1].reshape(1, 9), g["prev_rates"][k:k + 1, -1],
                                  g["prev_thrust"][k:k + 1, -1:]], axis=1)
            err = soa_vs_oracle(drone.state.cpu().numpy(), ref, 1)
            assert err["pos_comp"] < REL_TOL and err["quat_abs"] < REL_TOL, (k, err)
            RT, gyro, acc = (x.cpu().numpy()[0] for x in ret)
            np.testing.assert_allclose(RT, g["ret_RT"][k], atol=2e-5)
    assert yaml.safe_dump(params) == frozen, "Drone(params) must not modify the caller's dict (the reference does, :143-144)"

    g = load_golden("g2_sin_4096")
    for k in (0, 7):
        drone = Drone(params, num_envs=1, device=DEV)
        drone.reset(position=np.array(params["drone"]["initial_position"]), velocity=np.array(params["drone"]["initial_velocity"]),
                    ypr=np.array(params["drone"]["initial_orientation"]))                           # simulator.py:59
        for i in range(g["actions"].shape[0]):
            ret = drone.step(action=g["actions"][i, k], wind_velocity_vector=np.array([0, 0, 0]), object_list=[])
            assert not bool(drone.done)                                                              # :92-94
        ref = np.concatenate([g["state"][k:k + 1, -1], g["R"][k:k + 1, -1].reshape(1, 9), g["prev_rates"][k:k + 1, -1],
                              g["prev_thrust"][k:k + 1, -1:]], axis=1)
        assert_parity(soa_vs_oracle(drone.state.cpu().numpy(), ref, 1), REL_TOL, f"simulator sequence, G2 drone {k}")
        np.testing.assert_allclose(ret[0].cpu().numpy()[0], g["ret_RT"][k], atol=1e-5)
        np.testing.assert_allclose(drone.position.cpu().numpy()[0], g["state"][k, -1, 0:3], rtol=1e-5, atol=1e-5)
    drone2 = Drone(DEFAULT_PARAMS_PATH, num_envs=3, device=DEV)         # a YAML path works too
    assert drone2.dt == pytest.approx(1 / 60) and drone2.max_rates == 200


def test_set_done_bits_target_public_api(params_1k):
    n, k = 1000, 40
    words = (n + 63) // 64
    env = _drone_batch(params_1k.replace(ceiling=10.02), n, with_done_bits=True, auto_reset=True)
    env.reset()
    acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=2)).to(DEV)
    acts[..., 3] = 1.0                                                   # full throttle: the ceiling ends episodes
    rows = torch.zeros((k, words), dtype=torch.int64, device=DEV)
    dones = torch.zeros((k, n), dtype=torch.uint8, device=DEV)
    for t in range(k):                                                   # per-step destinations (what bench.py's gather does)
        env.set_done_bits_target(rows[t])
        env.step(acts[t], return_imu=False)
        dones[t] = env.done_u8
    env.set_done_bits_target(None)
    env.step(acts[0], return_imu=False)
    torch.cuda.synchronize()
    from fpyv_amd.dist import unpack_done_bits
    for t in range(k):
        assert torch.equal(unpack_done_bits(rows[t], n), dones[t])
    assert bool(dones.any())
    assert torch.equal(unpack_done_bits(env.done_bits, n), env.done_u8)
    with pytest.raises(ValueError):
        env.set_done_bits_target(torch.zeros(words - 1, dtype=torch.int64, device=DEV))
    with pytest.raises(ValueError):
        env.set_done_bits_target(rows, stride_words=words - 1)


def test_c_abi_allgather_done_over_rccl(params_1k):
    """fpv_comm_* / fpv_allgather_done: the done-mask exchange for a non-Python host, through RCCL opened
    at run time.  One GPU here, so a communicator of one rank (the driver's multi-GPU run covers N > 1):
    the gathered block must be this rank's masks, for a single mask and for a [k, words] bucket written
    by the k-step kernel, and the fp32 variant must carry the episode returns."""
    import ctypes as C
    L = _lib.lib()
    ident = (C.c_uint8 * _lib.FPV_COMM_ID_BYTES)()
    _lib.check(L.fpv_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(L.fpv_comm_create(ident, 1, 0, 0, C.byref(comm)))
    try:
        n, k = 5000, 16
        words = (n + 63) // 64
        env = _drone_batch(params_1k.replace(ceiling=10.01), n, with_done_bits=True, auto_reset=True, track_episodes=True)
        env.reset()
        acts = torch.from_numpy(sticks.ema_noise(k, range(n), seed=4)).to(DEV)
        acts[..., 3] = 1.0
        bucket = torch.zeros((k, words), dtype=torch.int64, device=DEV)
        env.set_done_bits_target(bucket, stride_words=words)
        env.rollout(acts)                                              # one launch writes all k mask rows
        gathered = torch.full((k, words), -1, dtype=torch.int64, device=DEV)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.fpv_allgather_done(comm, bucket.data_ptr(), gathered.data_ptr(), k * words, stream))
        returns = torch.zeros(n, device=DEV)
        _lib.check(L.fpv_allgather_f32(comm, env.last_return.data_ptr(), returns.data_ptr(), n, stream))
        torch.cuda.synchronize()
        assert torch.equal(gathered, bucket) and bool((bucket != 0).any())
        assert torch.equal(returns, env.last_return)
        assert L.fpv_allgather_done(comm, None, gathered.data_ptr(), words, stream) == -1
        assert L.fpv_allgather_done(None, bucket.data_ptr(), gathered.data_ptr(), words, stream) == -1
        # what a benchmark line certifies itself with: world size and rank of the communicator, the RCCL actually loaded
        ws, rk, ver = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.check(L.fpv_comm_info(comm, C.byref(ws), C.byref(rk), C.byref(ver)))
        assert (ws.value, rk.value) == (1, 0) and ver.value >= 20000, (ws.value, rk.value, ver.value)
        assert L.fpv_comm_info(None, None, None, None) == -1
        # the handle's 64-bit step counter through the C ABI
        cnt = C.c_uint64(0)
        _lib.check(L.fpv_get_step_counter(env._handle, C.byref(cnt)))
        assert cnt.value == k
        _lib.check(L.fpv_set_step_counter(env._handle, (1 << 40) + 7))
        _lib.check(L.fpv_get_step_counter(env._handle, C.byref(cnt)))
        assert cnt.value == (1 << 40) + 7
        # fpv_step_n reads action rows only (ABI 4): SoA sticks are refused with a message, fpv_step takes them
        b = _lib.FpvBuffers()
        C.memmove(C.byref(b), C.byref(env._buf), C.sizeof(b))
        soa = torch.zeros((4, env.ld), device=DEV)
        b.action, b.action_ld = soa.data_ptr(), env.ld
        assert L.fpv_step_n(env._handle, C.byref(b), 2, 0, 0, stream) == -1 and b"action rows" in L.fpv_last_error()
        assert L.fpv_step(env._handle, C.byref(b), stream) == 0
        torch.cuda.synchronize()
    finally:
        L.fpv_comm_destroy(comm)
    bad = C.c_void_p()
    assert L.fpv_comm_create(ident, 2, 5, 0, C.byref(bad)) == -1 and not bad.value


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason=f"needs two GPUs in one process (this box shows {torch.cuda.device_count()}): wakes up on the driver's multi-GPU node")
@pytest.mark.timeout(600)
def test_two_gpus_one_process_two_handles_rccl_world_size_2(params_1k):
    """VERDICT r3 #8 / SURVEY 8b "one process with 8 handles", at the smallest size that exercises it: a handle and a
    communicator rank per GPU (devices 0 and 1) driven by two host threads of one process.  Every fpv_* call must bind its
    handle's device for its own launches and put the caller's back (DeviceGuard, fpv_hip.hip: rank 1's thread has device
    0 current throughout); fpv_comm_create at world size 2 is this project's first RCCL communicator with more than one
    rank; fpv_allgather_done ships a [k, words] bucket the k-step kernel filled (one mask row per step) and every rank
    must receive the concatenation of both ranks' buckets."""
    _two_host_threads_two_handles(params_1k, devices=[0, 1], one_world=True)


@pytest.mark.timeout(600)
def test_two_host_threads_two_handles_on_one_gpu(params_1k):
    """The same program with both handles on GPU 0 and a one-rank communicator per thread (RCCL refuses two ranks on one
    device): what of the two-GPU test a one-GPU box can run - concurrent fpv_* calls from two threads, thread-local error
    strings, RCCL opened once under call_once, the k-step bucket through fpv_allgather_done."""
    _two_host_threads_two_handles(params_1k, devices=[0, 0], one_world=False)


def test_integration_md_stub_runs_and_lands_on_the_reference(params_1k):
    """The binding INTEGRATION.md shows a reference maintainer (src/utils/hip_drone.py) is executed as
    written: Drone(params dict), reset, 1000 x step with the reference's arguments, against capture G2."""
    import os
    import re
    import yaml
    from conftest import REPO
    from fpyv_amd.params import DEFAULT_PARAMS_PATH
    txt = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# src/utils/hip_drone\.py.*?)```", txt, re.S).group(1)
    assert "..." not in code, "the stub must be complete"
    ns = {}
    exec(compile(code, "hip_drone.py", "exec"), ns)
    with open(DEFAULT_PARAMS_PATH) as f:
        params = yaml.safe_load(f)
    params["simulator"]["fps"] = 1000
    g = load_golden("g2_sin_4096")
    k = 3
    drone = ns["Drone"](params, num_envs=1, device=DEV)
    drone.reset(position=np.array(params["drone"]["initial_position"]), velocity=np.array(params["drone"]["initial_velocity"]),
                ypr=np.array(params["drone"]["initial_orientation"]))
    for i in range(g["actions"].shape[0]):
        RT, gyro, acc = drone.step(g["actions"][i, k], np.array([0, 0, 0]), [])
    torch.cuda.synchronize()
    ref = np.concatenate([g["state"][k:k + 1, -1], g["R"][k:k + 1, -1].reshape(1, 9), g["prev_rates"][k:k + 1, -1],
                          g["prev_thrust"][k:k + 1, -1:]], axis=1)
    assert_parity(soa_vs_oracle(drone.state.cpu().numpy(), ref, 1), REL_TOL, "INTEGRATION.md stub")
    np.testing.assert_allclose(RT.cpu().numpy()[0], g["ret_RT"][k], atol=1e-5)
    np.testing.assert_allclose(acc.cpu().numpy()[0], g["accel"][k, -1], rtol=1e-4, atol=1e-4)
    assert not bool(drone.done)


def test_plain_c_host_through_the_c_abi(params_1k, tmp_path):
    """The boundary is a C ABI: examples/c_host/main.c - plain C, hipMalloc'd buffers, no Python or torch in
    the process - is compiled here with gcc against include/fpv_abi.h and libfpv_hip.so, runs config 2
    (4096 drones, sinusoidal sticks) as k fpv_step launches, as one fpv_step_n launch, and as TWO handles over the column
    halves of the same buffers on two streams (the split-phase layout against the bare C ABI), and must reproduce the
    Python host's result bit for bit every time."""
    import ctypes as C
    import os
    import subprocess
    from conftest import REPO
    n, k = 4096, 200
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(REPO, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(REPO, "examples", "c_host", "main.c"), "-L", os.path.join(REPO, "fpyv_amd"), "-lfpv_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(REPO, "fpyv_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    acts = sticks.sinusoid(k, n, params_1k.dt)
    cp = _lib.pack_params(params_1k)
    (tmp_path / "params.bin").write_bytes(bytes(C.string_at(C.addressof(cp), C.sizeof(cp))))
    (tmp_path / "actions.bin").write_bytes(acts.tobytes())
    env = _drone_batch(params_1k, n, with_accel=False)
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV), fused=False)
    torch.cuda.synchronize()
    want = env.state.cpu().numpy()
    rows, ld = want.shape
    for mode in ("steps", "fused", "split"):
        out = tmp_path / f"state_{mode}.bin"
        r = subprocess.run([exe, str(tmp_path / "params.bin"), str(tmp_path / "actions.bin"), str(n), str(k), str(out)]
                           + ([mode] if mode != "steps" else []), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        raw = np.fromfile(out, dtype=np.uint8)
        got = raw[:rows * ld * 4].view(np.float32).reshape(rows, ld)
        rew = raw[rows * ld * 4:rows * ld * 4 + n * 4].view(np.float32)
        done = raw[rows * ld * 4 + n * 4:]
        assert np.array_equal(got[:, :n].view(np.uint32), want[:, :n].view(np.uint32)), mode
        assert np.array_equal(rew.view(np.uint32), env.reward.cpu().numpy().view(np.uint32)) and not done.any()


def test_gravity_force_helper_as_written(params_1k):
    """Drone.get_gravity_force_in_drone_ref_frame = R @ [0, 0, -9.81 m] (components.py:254-255), R body -> world."""
    env = _drone_batch(params_1k, 5)
    env.reset(ypr=np.array([[0, 0, 0], [30, 0, 0], [0, 45, 0], [10, -20, 70], [180, 0, 0]], dtype=np.float32))
    got = env.get_gravity_force_in_drone_ref_frame().cpu().numpy()
    R = env.rotation_matrix.cpu().numpy().astype(np.float64)
    want = R @ np.array([0, 0, -9.81 * params_1k.mass])
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[0], [0, 0, -9.81 * 0.75], atol=1e-6)
    np.testing.assert_allclose(got[4], [0, 0, 9.81 * 0.75], atol=1e-5)      # rolled upside down


def test_reference_scalar_attributes_of_drone(params_1k):
    """The attributes simulator.py reads off the Drone besides position / velocity / done (`throttle` :161,
    `prev_rates` / `prev_thrust` :64-65) and the thrust-curve helpers of components.py:136-142, against the
    constants captured from the reference."""
    g = load_golden("params_golden")
    env = _drone_batch(params_1k, 7)
    env.reset()
    assert env.throttle is None
    a = torch.linspace(-1, 1, 28, device=DEV).reshape(7, 4).contiguous()
    env.step(a, return_imu=False)
    assert torch.equal(env.throttle, a[:, 3])
    env.step(np.array([0.1, 0.2, 0.3, -0.25], dtype=np.float32), return_imu=False)      # broadcast sticks
    assert torch.allclose(env.throttle, torch.full((7,), -0.25, device=DEV))
    np.testing.assert_allclose(env.throttle2thrust(g["stick_samples"]), g["thrust_samples"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(env.thrust2throttle(np.array([0.0, 5.0, 31.5, 60.0, 90.0])), g["thrust2throttle_samples"], rtol=1e-10, atol=1e-12)
    assert abs(env.min_throttle_in_force - float(g["min_throttle_in_force"])) < 1e-10
    assert abs(env.max_throttle_in_force - float(g["max_throttle_in_force"])) < 1e-10
    assert env.mass == float(g["mass"]) and env.gravity == float(g["gravity"]) and env.max_rates == float(g["max_rates"])
    assert env.prev_rates.shape == (7, 3) and env.prev_thrust.shape == (7,)
    # after a rollout `throttle` reports the LAST step's sticks of that rollout, not the step() before it
    torch.manual_seed(12)
    acts = torch.rand((5, 7, 4), device=DEV) * 2 - 1
    env.rollout(acts)
    assert torch.equal(env.throttle, acts[-1, :, 3])
    # step(A); rollout(...); step(A) with the SAME tensor object: the in-place-policy fast path of _action_ptr must report A
    # again (it used to leave `throttle` on the rollout's batch), and a tensor re-shaped in place is validated again
    env.step(a, return_imu=False)
    env.rollout(acts)
    env.step(a, return_imu=False)
    assert torch.equal(env.throttle, a[:, 3])
    a.resize_(8, 4)
    with pytest.raises(ValueError):
        env.step(a, return_imu=False)


def test_force_multiplier_pid_built_and_reset_like_the_reference(params_1k):
    """Drone.force_multiplier_pid (components.py:143-145): PID(**params['drone']['force_multiplier_pid'], dt=dt) with
    min_output / max_output REPLACED by the 5 %-throttle and full-throttle forces, reset by Drone.reset (:166).  One
    controller per drone; its arithmetic is the a16 kernel (checked against the reference class in
    test_components_pid_kernel_vs_reference_class), here: construction constants, per-drone targets, reset."""
    from fpyv_amd.components import Drone
    from fpyv_amd.params import DEFAULT_PARAMS_PATH
    import yaml
    cfg = yaml.safe_load(open(DEFAULT_PARAMS_PATH))
    cfg["simulator"]["fps"] = 1000
    before = yaml.safe_dump(cfg)
    d = Drone(cfg, num_envs=5, device=DEV)
    assert yaml.safe_dump(cfg) == before, "the caller's params dict must not be modified (the reference mutates it)"
    pid = d.force_multiplier_pid
    g = load_golden("params_golden")
    assert (pid.kP, pid.kI, pid.kD) == (0.1, 2.0, 0.05) and pid.integral_clip == 100.0 and pid.derivative_transition_rate == 0.2
    assert abs(pid.min_output - float(g["min_throttle_in_force"])) < 1e-10      # components.py:143
    assert abs(pid.max_output - float(g["max_throttle_in_force"])) < 1e-10      # components.py:144
    assert pid.dt == d.dt == 1e-3 and pid.n == 5
    # the call of components.py:288: multiplier = pid(measured_dist2target, keep_distance); numpy arrays of targets work
    dist = np.array([3.0, 7.0, 9.0, 12.0, 30.0], dtype=np.float32)
    out1 = pid(dist, 6.0).clone()
    out2 = pid(dist, np.full(5, 6.0)).clone()                 # per-drone targets as an ndarray
    out3 = pid(torch.from_numpy(dist).to(DEV), [6.0] * 5).clone()
    torch.cuda.synchronize()
    want, _ = lane_model.pid_run([0.1, 2.0, 0.05, 1e-3, 100.0, pid.min_output, pid.max_output, 0.2], np.repeat(dist[2:3], 3), np.full(3, 6.0))
    assert np.array_equal(np.array([out1[2].item(), out2[2].item(), out3[2].item()], dtype=np.float32).view(np.uint32), want.view(np.uint32))
    assert float(out1.min()) >= pid.min_output - 1e-6 and float(out3.max()) <= pid.max_output + 1e-6
    assert float(pid.integral.abs().max()) > 0
    d.reset(mask=np.array([1, 0, 1, 0, 1], dtype=np.uint8))                   # components.py:166, masked like the drones
    torch.cuda.synchronize()
    integ = pid.integral.cpu().numpy()
    assert np.all(integ[[0, 2, 4]] == 0) and np.all(integ[[1, 3]] != 0)
    d.reset()
    torch.cuda.synchronize()
    assert float(pid.integral.abs().max()) == 0.0 and bool((pid.state[3, :5] == 1).all())
    with pytest.raises(ValueError):
        pid(np.zeros(4), 1.0)


def test_failed_step_does_not_leak_the_guidance_override(params_1k):
    """a step that raises after rotation_matrix= was bound (bad object row, too many objects) must not
    leave the matrix in place for the next plain step."""
    n = 9
    env, ref = _drone_batch(params_1k, n), _drone_batch(params_1k, n)
    env.reset(); ref.reset()
    a = torch.zeros((n, 4), device=DEV)
    R = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float32)
    with pytest.raises((TypeError, ValueError)):
        env.step(a, object_list=[object()], rotation_matrix=R, thrust_force=5.0, return_imu=False)
    with pytest.raises((TypeError, ValueError)):
        env.step(a, object_list=[(0, 0, 0, 0, 0, 0)] * 9, rotation_matrix=R, thrust_force=5.0, return_imu=False)
    assert not env._buf.rotation_override and not env._buf.thrust_override
    env.step(a, return_imu=False); ref.step(a, return_imu=False)
    env.rollout(a, steps=3); ref.rollout(a, steps=3)                          # "use fpv_step" if the override had leaked
    torch.cuda.synchronize()
    assert torch.equal(env.state, ref.state)


def test_calls_restore_the_callers_current_device(params_1k):
    """SURVEY 8b "one process with 8 handles": an fpv_* call binds the handle's device for its own launches and puts
    the caller's current device back.  With one GPU the observable part is that the current device is never left
    changed and that a handle created for device 0 works from any thread state; the guard itself (DeviceGuard,
    fpv_hip.hip) is what a multi-GPU host relies on."""
    import ctypes as C
    L = _lib.lib()
    before = torch.cuda.current_device()
    env = _drone_batch(params_1k, 128)
    env.reset()
    env.step(torch.zeros((128, 4), device=DEV), return_imu=False)
    env.rollout(torch.zeros((4, 128, 4), device=DEV))
    from fpyv_amd.components import PID
    PID(1, 0, 0, 1e-3, num_envs=4, device=DEV)(torch.zeros(4, device=DEV), 0.0)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    if torch.cuda.device_count() > 1:                 # the driver's 8-GPU box: a handle on GPU 1 driven while GPU 0 is current
        other = _lib.pack_params(params_1k)
        h = C.c_void_p()
        assert L.fpv_create(C.byref(other), 64, 1, C.byref(h)) == 0
        st = torch.zeros((14, 64 + 256), device="cuda:1")
        b = _lib.FpvBuffers()
        b.state, b.ld = st.data_ptr(), st.shape[1]
        torch.cuda.set_device(0)
        assert L.fpv_reset(h, C.byref(b), None, None, None, None, None) == 0
        assert torch.cuda.current_device() == 0
        torch.cuda.synchronize(1)
        assert float(st[2, 0]) == 10.0
        L.fpv_destroy(h)


@pytest.mark.parametrize("api", ["step", "rollout"])
def test_two_ranks_rehearsed_on_one_gpu(params_1k, tmp_path, api):
    """The real N-rank path of bench.py at world size 2 - self-launched ranks, per-rank stick streams, the step kernels
    (api=step) or the k-step kernel writing one mask row per step into the bucket (api=rollout: done_bits_stride), the
    bucketed asynchronous done-mask all-gather with its FLUSH of a partly filled last bucket, MAX over ranks, one JSON
    line - with both ranks on GPU 0 over gloo (RCCL refuses two ranks on one device; the driver's multi-GPU run uses
    RCCL).  The ceiling sits 0.5 mm above the start height with auto-reset on, so done bits ARE set on most steps
    (with a 100 m ceiling every mask was zero and a kernel that never wrote the rows would have passed), and
    warm-up + steps = 45 is not a multiple of the 16-step bucket, so the last bucket travels through flush() with 13
    rows.  Every rank's final state and the gathered masks must equal a single-process run of the same shards."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    n, steps, warm, ring, ceiling = 4096, 37, 8, 8, 10.0005
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", str(steps),
                        "--warmup", str(warm), "--drones-per-gpu", str(n), "--ring", str(ring), "--preheat-s", "0", "--no-cpu-baseline",
                        "--ceiling", str(ceiling), "--api", api, "--dump-gathered", str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["data"].startswith("rehearsal")
    assert out["config"]["global_drones"] == 2 * n and "allgather(done_bits x16 steps)" in out["config"]["parallelism"]
    assert out["collective"]["world_seen"] == 2 and out["collective"]["backend"] == "gloo"
    gathered = np.load(tmp_path / "gathered_last_bucket.npy")              # [world, rows, words]
    total, block = warm + steps, 16
    first = (total - 1) // block * block
    assert total % block != 0 and gathered.shape == (2, total - first, n // 64), "the last bucket is a flushed, partly filled one"
    assert gathered.any(), "the scenario must set done bits"
    p = load_params(fps=1000, ceiling=ceiling)
    for rank in range(2):
        acts = sticks.ema_noise_device(ring, n, DEV, seed=1234 + rank)
        ref = _drone_batch(p, n, auto_reset=True, with_accel=False, with_done_bits=True)
        ref.reset()
        set_rows = 0
        for t in range(total):
            ref.step(acts[t % ring], return_imu=False)
            if t >= first:
                bits = ref.done_bits.cpu().numpy()
                set_rows += int(bits.any())
                assert np.array_equal(bits, gathered[rank, t - first]), f"rank {rank} step {t}"
        assert set_rows >= 3, "several of the flushed rows must carry set bits"
        assert np.array_equal(ref.state.cpu().numpy().view(np.uint32), np.load(tmp_path / f"state_rank{rank}.npy").view(np.uint32))


def test_sharded_example_under_the_launcher_two_ranks_on_one_gpu():
    """examples/sharded_vec_env.py - a population cut into contiguous shards, in-kernel sticks keyed by the global drone id,
    the k-step kernel writing one mask row per step into DoneGather's bucket - under `python -m torch.distributed.run` with two
    ranks on GPU 0 over gloo: the gathered masks agree with the ranks' own done flags and rank 0's shard equals its slice of
    the unsharded run bit for bit (the script asserts both)."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "examples", "sharded_vec_env.py"), "--drones", "65536", "--steps", "150",
                        "--block", "64", "--backend", "gloo", "--all-ranks-on-gpu0", "--check"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "2 ranks x 32768 drones over gloo, 150 steps" in r.stdout and "(must agree)" in r.stdout
    assert "equals its slice of the unsharded run bit for bit" in r.stdout


@pytest.mark.parametrize("script,args,expect", [
    ("simulator_headless.py", ["--drones", "2048", "--steps", "300"], ["2048 drones x 300 steps", "crashed into ground/obstacles/target"]),
    ("guidance_headless.py", ["--drones", "2048", "--steps", "300"], ["2048 drones x 300 guided steps", "closest approach to the moving target"]),
    ("closed_loop_policy.py", ["--drones", "16384", "--steps", "60", "--partitions", "1", "2"], ["16384 drones, 60 closed-loop steps", "split phase with 2 partitions"]),
], ids=["simulator", "guidance", "closed_loop"])
def test_examples_run_end_to_end(script, args, expect):
    """The headless replays of the reference's own loops (examples/: simulator.py:83-156 with its object list, the guidance branch of
    :110, the closed policy loop on the zero-copy observation) run to the end against the library as it is built now."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", script)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    for s in expect:
        assert s in r.stdout, (s, r.stdout[-2000:])
