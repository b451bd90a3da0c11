"""In-kernel stick-noise generator: Philox4x32 pinned by the Random123 known-answer vectors for
10 and for 7 rounds (the generator runs 7), the table-driven inverse normal CDF against scipy's, the EMA profile of
/root/reference/tests/noise_smooth_test.py:6-12 against a float64 NumPy restatement (oracle/philox.py) and against
its own statistics (mean, stationary sigma, lag-1 autocorrelation, normality), shard / batch invariance, and - on the
GPU - the kernel against both."""
import numpy as np
import pytest

from fpyv_amd import load_params
from oracle import lane_model, philox

KAT = [  # Random123 kat_vectors, philox4x32-10: counter, key, expected
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


KAT7 = [  # Random123 kat_vectors, philox4x32-7 (seven rounds: the "Crush-resistant" minimum of the Random123 paper)
    ([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
]


@pytest.mark.parametrize("rounds,ctr,key,want", [(10,) + k for k in KAT] + [(7,) + k for k in KAT7])
def test_philox_known_answers(rounds, ctr, key, want):
    got_np = philox.philox4x32(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32), rounds)[0]
    assert [int(x) for x in got_np] == want
    assert lane_model.philox(ctr, key, rounds) == want          # the header the kernel compiles
    assert lane_model.noise_philox_rounds() == philox.NOISE_ROUNDS == 7


def test_ema_profile_matches_float64_restatement():
    p = load_params(fps=1000)
    n, steps, seed, off = 300, 400, 0x1234_5678_9abc_def0, (1 << 33) + 17
    applied, ns = lane_model.stick_noise(p, n, steps, noise_seed=seed, drone_id_offset=off)
    ref, ref_s = philox.ema_sticks(seed, off + np.arange(n, dtype=np.uint64), steps)
    assert np.abs(applied - ref).max() < 3e-6
    assert np.abs(ns[:, :n].T - ref_s).max() < 3e-6
    # the profile itself: zero mean, stationary sigma = sqrt(tau / (2 - tau)) = 0.2294 (noise_smooth_test.py)
    tail = applied[200:].reshape(-1)
    assert abs(tail.mean()) < 5e-3 and abs(tail.std() - np.sqrt(0.1 / 1.9)) < 5e-3
    assert np.abs(applied).max() <= 1.0


def test_inverse_normal_table_against_scipy():
    """fpv_normal_from_word (one table row + a cubic, no log / sqrt / division) against the exact inverse CDF in float64:
    every binade of the tail probability down to the smallest one the generator produces, every table-row boundary from
    both sides, the words that round up to p = 0.5, both signs, and a dense random sample.  Max |error| 3e-6; the
    generated header records 2.2e-6."""
    rng = np.random.default_rng(0)
    edges = []
    for e in range(31):
        for j in range(4):
            k0 = int((1 << e) * (1 + j / 4))
            edges += [k0 - 1, k0, k0 + 1, k0 + 2]
    edges += [0, 1, 2, 3, (1 << 31) - 1, (1 << 31) - 2, (1 << 31) - 64, (1 << 31) - 65, (1 << 31) - 129]
    k = np.concatenate([np.array([x for x in edges if 0 <= x < (1 << 31)], dtype=np.uint32),
                        rng.integers(0, 1 << 31, 200000).astype(np.uint32),
                        (rng.integers(0, 1 << 31, 50000) >> rng.integers(0, 31, 50000)).astype(np.uint32)])     # every binade well populated
    for sign in (0, 1):
        w = k | np.uint32(sign << 31)
        z, ref = lane_model.normal_from_words(w), philox.normal_from_words(w)
        assert np.abs(z - ref).max() < 3e-6, np.abs(z - ref).max()
        assert np.all(np.signbit(z) == bool(sign))
    # symmetric to the bit, monotone in the tail probability, bounded by the smallest p = 2^-32
    zp, zm = lane_model.normal_from_words(k), lane_model.normal_from_words(k | np.uint32(1 << 31))
    assert np.array_equal(zp, -zm)
    ks = np.sort(np.unique(k | np.uint32(1)))
    zs = lane_model.normal_from_words(ks).astype(np.float64)
    assert np.all(np.diff(zs) <= 5e-6), "|z| falls as the tail probability grows (up to the table's own error)"
    assert 6.22 < float(lane_model.normal_from_words(np.array([0], dtype=np.uint32))[0]) < 6.24      # -Phi^-1(2^-32) = 6.2303


def test_noise_profile_statistics():
    """The statistics that define the profile (noise_smooth_test.py:6-12: x ~ N(0, 1), x_s <- 0.9 x_s + 0.1 x): the
    driving normals are normal (moments and a Kolmogorov-Smirnov test on 4 x 10^5 draws, independent across channels,
    drones and steps), the smoothed sticks have zero mean, the stationary sigma sqrt(tau / (2 - tau)) = 0.2294 and lag-1
    autocorrelation 1 - tau = 0.9."""
    from scipy import stats
    p = load_params(fps=1000)
    n, steps = 400, 1000
    applied, _ = lane_model.stick_noise(p, n, steps, noise_seed=2024, drone_id_offset=12345)
    x = applied[300:].astype(np.float64)                               # [T, n, 4], stationary part
    assert abs(x.mean()) < 3e-3
    assert abs(x.std() - np.sqrt(0.1 / 1.9)) < 3e-3
    a, b = x[:-1] - x.mean(), x[1:] - x.mean()
    assert abs((a * b).mean() / x.var() - 0.9) < 5e-3
    # the driving normals, recovered from the recurrence: z = (x_s(t) - 0.9 x_s(t-1)) / 0.1 (fp32 sticks: ~1e-6 of noise)
    z = ((applied[1:].astype(np.float64) - 0.9 * applied[:-1].astype(np.float64)) / 0.1)[:250].reshape(-1)
    assert z.size == 400000 and abs(z.mean()) < 6e-3 and abs(z.std() - 1) < 5e-3
    assert abs(stats.skew(z)) < 0.02 and abs(stats.kurtosis(z)) < 0.04
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # no correlation between channels, neighbouring drones or consecutive steps of the driving noise
    zz = ((applied[1:].astype(np.float64) - 0.9 * applied[:-1].astype(np.float64)) / 0.1)
    for u, v in ((zz[..., 0], zz[..., 1]), (zz[:, :-1, 2], zz[:, 1:, 2]), (zz[:-1, :, 3], zz[1:, :, 3])):
        assert abs(np.corrcoef(u.reshape(-1), v.reshape(-1))[0, 1]) < 6e-3


def test_streams_depend_only_on_global_id_and_step():
    p = load_params(fps=1000)
    whole, _ = lane_model.stick_noise(p, 256, 50, noise_seed=9)
    shard, _ = lane_model.stick_noise(p, 128, 50, noise_seed=9, drone_id_offset=128)
    assert np.array_equal(whole[:, 128:], shard), "a shard must reproduce its slice of the global batch"
    # resuming at step 20 with the saved EMA state continues the same stream
    first, ns = lane_model.stick_noise(p, 64, 20, noise_seed=9)
    rest, _ = lane_model.stick_noise(p, 64, 30, noise_seed=9, step0=20, ns=ns)
    assert np.array_equal(np.concatenate([first, rest]), whole[:, :64])
    other, _ = lane_model.stick_noise(p, 64, 50, noise_seed=10)
    assert not np.array_equal(other, whole[:, :64])


def test_step_index_is_64_bit_and_does_not_wrap():
    """The Philox counter carries the step index in words 2 AND 3 (since ABI 4): the stream continues across 2^32 steps
    (5.5 h at the k-step kernel's rate) instead of repeating."""
    p = load_params(fps=1000)
    n, seed, off = 96, 0xfeed_f00d_1234, 7
    ids = off + np.arange(n, dtype=np.uint64)
    edge = (1 << 32) - 3
    # the integer generator itself: counter word 3 = high word of the step
    for step in (edge + 2, 1 << 32, (1 << 32) + 5, (123 << 32) | 77):
        want = philox.philox4x32(np.array([[off & 0xFFFFFFFF, off >> 32, step & 0xFFFFFFFF, step >> 32]], dtype=np.uint32),
                                 np.array([[seed & 0xFFFFFFFF, seed >> 32]], dtype=np.uint32), 7)[0]
        assert lane_model.philox([off & 0xFFFFFFFF, off >> 32, step & 0xFFFFFFFF, step >> 32], [seed & 0xFFFFFFFF, seed >> 32], 7) == [int(x) for x in want]
    # six steps across the boundary: host build of the kernel's generator vs the float64 restatement with a 64-bit step
    across, ns = lane_model.stick_noise(p, n, 6, noise_seed=seed, drone_id_offset=off, step0=edge)
    ref, ref_s = philox.ema_sticks(seed, ids, 6, step0=edge)
    assert np.abs(across - ref).max() < 3e-6 and np.abs(ns[:, :n].T - ref_s).max() < 3e-6
    # a 32-bit counter would have wrapped: after steps 2^32-3 .. 2^32-1 it replayed steps 0, 1, 2 of the stream
    head, ns3 = lane_model.stick_noise(p, n, 3, noise_seed=seed, drone_id_offset=off, step0=edge)
    wrapped, _ = lane_model.stick_noise(p, n, 3, noise_seed=seed, drone_id_offset=off, step0=0, ns=ns3.copy())
    assert np.array_equal(head, across[:3]) and not np.array_equal(wrapped, across[3:])
    # resuming in the middle of the crossing with the saved EMA state continues the same stream bit for bit
    a, ns_a = lane_model.stick_noise(p, n, 2, noise_seed=seed, drone_id_offset=off, step0=edge)
    b, _ = lane_model.stick_noise(p, n, 4, noise_seed=seed, drone_id_offset=off, step0=edge + 2, ns=ns_a)
    assert np.array_equal(np.concatenate([a, b]), across)
    # the stochastic-rounding seed: base + step below 2^32 (= ABI <= 3), high word folded in beyond
    assert lane_model.round_seed(11, 5) == 16 and lane_model.round_seed(0xFFFFFFFF, 2) == 1
    assert lane_model.round_seed(11, (1 << 32) + 5) == (16 + 0x9e3779b1) & 0xFFFFFFFF
    assert lane_model.round_seed(11, (1 << 32) + 5) != lane_model.round_seed(11, 5)


def test_reduced_sincos_without_libm():
    """fpv_sincos_reduced (big-angle step, fp32 Racer, reset kernel): Cody-Waite by pi/2 + the short polynomials, the
    same instructions on the host and on gfx950.  Against float64 over the half-angles a step can produce."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-4, 4, 20000), rng.uniform(-60, 60, 20000), rng.uniform(-1e3, 1e3, 5000),
                        np.arange(-8, 9) * (np.pi / 4), np.arange(-8, 9) * (np.pi / 2), [0.0, 1e-8, -1e-8]]).astype(np.float32)
    s, c = lane_model.sincos_reduced(x)
    x64 = x.astype(np.float64)
    # the input is exact fp32; the reduction adds <= ulp(r) ~ 6e-8 to the 3e-9 of the polynomials
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7, np.abs(s - np.sin(x64)).max()
    assert np.abs(c - np.cos(x64)).max() < 1.5e-7, np.abs(c - np.cos(x64)).max()
    assert np.all(np.abs(s * s + c * c - 1) < 4e-7)


def test_noise_added_to_policy_action_and_clipped():
    p = load_params(fps=1000).replace(noise_gain=2.0)
    base = np.full((30, 16, 4), 0.9, dtype=np.float32)
    applied, _ = lane_model.stick_noise(p, 16, 30, noise_seed=1, base_actions=base)
    ref, _ = philox.ema_sticks(1, np.arange(16, dtype=np.uint64), 30, gain=2.0, base_action=base.astype(np.float64))
    assert np.abs(applied - ref).max() < 5e-6 and applied.max() == 1.0


@pytest.mark.gpu
def test_kernel_stick_noise_vs_references():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU: the stepper has no CPU path")
    from fpyv_amd.env import DroneBatch
    from oracle import oracle
    from parity import assert_parity, soa_vs_oracle
    p = load_params(fps=1000)
    n, steps, seed, off = 1000, 200, 77, 5000
    env = DroneBatch(p, n, device="cuda:0", stick_noise=True, noise_seed=seed, drone_id_offset=off,
                     with_action_out=True, with_accel=False)
    env.reset()
    applied = np.zeros((steps, n, 4), dtype=np.float32)
    for t in range(steps):
        env.step(None, return_imu=False)                  # pure noise sticks
        applied[t] = env.action_out.cpu().numpy()
    ref, ref_s = philox.ema_sticks(seed, off + np.arange(n, dtype=np.uint64), steps)
    assert np.abs(applied - ref).max() < 5e-6
    assert np.abs(env.noise_state[:, :n].t().cpu().numpy() - ref_s).max() < 5e-6
    host, _ = lane_model.stick_noise(p, n, steps, noise_seed=seed, drone_id_offset=off)
    assert np.array_equal(applied, host), "the generator calls no library: host build and kernel agree bit for bit"
    # physics driven by those sticks == oracle driven by the recorded applied actions
    st = oracle.drone_initial_state(n, p.init_position, p.init_velocity, [0, 0, 0])
    oracle.drone_run(p, st, applied.astype(np.float64), threads=0)
    assert_parity(soa_vs_oracle(env.state.cpu().numpy(), st, n), 1e-5, "noise-driven flight")
    # policy action + noise, via rollout; and reset clears the EMA state of the masked lanes
    env2 = DroneBatch(p.replace(noise_gain=0.5), n, device="cuda:0", stick_noise=True, noise_seed=seed,
                      with_action_out=True, with_accel=False)
    env2.reset()
    base = torch.full((10, n, 4), 0.2, device="cuda:0")
    env2.rollout(base)
    ref2, _ = philox.ema_sticks(seed, np.arange(n, dtype=np.uint64), 10, gain=0.5,
                                base_action=np.full((10, n, 4), 0.2))
    assert np.abs(env2.action_out.cpu().numpy() - ref2[-1]).max() < 5e-6
    mask = np.zeros(n, dtype=np.uint8); mask[::3] = 1
    env2.reset(mask=mask)
    ns = env2.noise_state[:, :n].cpu().numpy()
    assert np.all(ns[:, ::3] == 0) and np.all(ns[:, 1::3] != 0)
    env3 = DroneBatch(p, n, device="cuda:0", stick_noise=True, noise_seed=seed, with_accel=False)
    env3.reset()
    env3.rollout(None, steps=5)
    with pytest.raises(ValueError):
        DroneBatch(p, 8, device="cuda:0").step(None)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 333])
def test_kernel_step_index_crosses_2_to_the_32(n):
    """set the handle's counter to 2^32 - 3, run 6 steps single-step and fused (the k-step kernel adds t to
    a 64-bit base in SGPRs): both equal the host build of the generator bit for bit, and differ from what a wrapping
    32-bit counter would have produced.  Also the checkpoint round trip of a counter beyond 2^32."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU: the stepper has no CPU path")
    from fpyv_amd.env import DroneBatch
    p = load_params(fps=1000)
    seed, off, edge = 0xabcdef12345, 1000, (1 << 32) - 3
    host, host_ns = lane_model.stick_noise(p, n, 6, noise_seed=seed, drone_id_offset=off, step0=edge)
    kw = dict(device="cuda:0", stick_noise=True, noise_seed=seed, drone_id_offset=off, with_action_out=True, with_accel=False)
    single = DroneBatch(p, n, **kw)
    single.reset(); single.set_step_counter(edge)
    got = np.zeros((6, n, 4), dtype=np.float32)
    for t in range(6):
        single.step(None, return_imu=False)
        got[t] = single.action_out.cpu().numpy()
    assert np.array_equal(got, host)
    assert single.step_counter() == edge + 6 == single.state_dict()["step_counter"] and single.step_counter() > (1 << 32)
    fused = DroneBatch(p, n, **kw)
    fused.reset(); fused.set_step_counter(edge)
    fused.rollout(None, steps=6)
    torch.cuda.synchronize()
    assert torch.equal(fused.state, single.state) and torch.equal(fused.noise_state, single.noise_state)
    assert np.array_equal(fused.action_out.cpu().numpy(), host[-1])
    assert np.array_equal(fused.noise_state[:, :n].cpu().numpy(), host_ns[:, :n])
    # what ABI <= 3 did: the counter wrapped, so steps 3..5 replayed steps 0..2 of the stream
    wrapped = DroneBatch(p, n, **kw)
    wrapped.reset(); wrapped.set_step_counter(0)
    ns3 = lane_model.stick_noise(p, n, 3, noise_seed=seed, drone_id_offset=off, step0=edge)[1]      # EMA state after steps 2^32-3 .. 2^32-1
    wrapped.noise_state[:, :n] = torch.from_numpy(ns3[:, :n]).to("cuda:0")
    wrapped.rollout(None, steps=3)
    assert not np.array_equal(wrapped.action_out.cpu().numpy(), host[-1])
    # resume from a checkpoint taken beyond 2^32
    ck = single.state_dict()
    cont = DroneBatch(p, n, **kw)
    cont.reset(); cont.load_state_dict(ck)
    cont.rollout(None, steps=4); single.rollout(None, steps=4)
    assert torch.equal(cont.state, single.state) and cont.step_counter() == edge + 10
    with pytest.raises(ValueError):
        cont.set_step_counter(1 << 64)
