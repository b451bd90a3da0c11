"""GPU parity tests: the HIP path, called through the C ABI (fpyv_amd.env -> ctypes ->
libfpv_hip.so), against the float64 oracle on the golden stick profiles, against the reference's
own captured outputs, and - at BASELINE.json's full sizes - through size-independent properties.

Tolerance (BASELINE.json north_star): 1e-5 relative on position / quaternion after 1000 steps."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]
DEV = "cuda:0"


def _drone_batch(p, n, **kw):
    from fpyv_amd.env import DroneBatch
    return DroneBatch(p, n, device=DEV, **kw)


def _run_golden(p, g, per_step_calls=False):
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(p, n)
    env.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    a = torch.from_numpy(acts).to(DEV)
    if per_step_calls:
        for t in range(T):
            env.step(a[t], wind_velocity_vector=g["wind"], object_list=[], return_imu=False)
    else:
        env.rollout(a, wind=g["wind"])
    torch.cuda.synchronize()
    return env


@pytest.mark.parametrize("name", ["g2_sin_4096", "g3_ema_noise", "g4_saturated", "g5_attitude_wind"])
def test_golden_profiles_vs_oracle_and_reference(params_1k, name):
    g = load_golden(name)
    n = g["actions"].shape[1]
    env = _run_golden(params_1k, g)
    got = env.state.cpu().numpy()
    ref = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    _, ref_acc, ref_done = oracle.drone_run(params_1k, ref, g["actions"].astype(np.float64), wind=g["wind"])
    assert_parity(soa_vs_oracle(got, ref, n), REL_TOL, name)
    # and directly against what the reference itself produced (last snapshot of the golden file)
    ref_direct = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1],
                                 g["prev_thrust"][:, -1:]], axis=1)
    assert_parity(soa_vs_oracle(got, ref_direct, n), REL_TOL, name + " (reference capture)")
    assert np.array_equal(env.done_u8.cpu().numpy(), g["done"][:, -1])
    # R_new @ acc is an OUTPUT (not state): thrust/m (up to 108 m/s^2), gravity and drag cancel in fp32, so it
    # carries ~6e-5 of its own magnitude (measured worst case, G4: 2.6e-4 m/s^2 on 75 m/s^2); 1e-4 relative + floor
    np.testing.assert_allclose(env.accel[:, :n].t().cpu().numpy(), g["accel"][:, -1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", range(4))
def test_other_drone_types_vs_reference_capture(k):
    """Captures G14 on the GPU: four drone types with every constructor parameter away from params.yaml (mass, drag,
    frame, max_rates 90-1200 deg/s, transition rates, motor block, fps 120-2000, gravity 1.62-9.81).  Single-step and
    k-step kernels, 1e-5 against the oracle and against the reference's own numbers, bit-identical to the host build."""
    from conftest import params_for_golden
    g = load_golden(f"g14_drone_type_{k}")
    p = params_for_golden(g)
    n = g["actions"].shape[1]
    ref = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    oracle.drone_run(p, ref, g["actions"].astype(np.float64), wind=g["wind"])
    ref_direct = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1],
                                 g["prev_thrust"][:, -1:]], axis=1)
    model = lane_model.initial_state(p, n, g["init_position"], g["init_velocity"], g["init_ypr"], as_reset_kernel=True)
    start = model.copy()
    lane_model.run(p, model, g["actions"], wind=g["wind"])
    fresh = _drone_batch(p, n)
    fresh.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    assert np.array_equal(fresh.state.cpu().numpy()[:, :n], start[:, :n]), "reset kernel != host build of its arithmetic"
    for per_step in (False, True):
        env = _run_golden(p, g, per_step_calls=per_step)
        got = env.state.cpu().numpy()
        assert_parity(soa_vs_oracle(got, ref, n), REL_TOL, f"g14 type {k}")
        assert_parity(soa_vs_oracle(got, ref_direct, n), REL_TOL, f"g14 type {k} (reference capture)")
        assert np.array_equal(env.done_u8.cpu().numpy(), g["done"][:, -1])
        assert np.array_equal(got[:, :n], model[:, :n]), "kernel and host build of the same arithmetic must agree bit for bit"


def test_reset_kernel_equals_host_build(params_1k):
    """fpv_reset with per-drone position / velocity / ypr (angles out to +-720 degrees) and a mask: every state row equals
    the host build of the same instructions bit for bit; unmasked drones keep their state."""
    rng = np.random.default_rng(12)
    n = 777
    pos, vel = rng.uniform(-50, 50, (n, 3)).astype(np.float32), rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    ypr = rng.uniform(-720, 720, (n, 3)).astype(np.float32)
    env = _drone_batch(params_1k, n)
    env.reset(position=pos, velocity=vel, ypr=ypr)
    torch.cuda.synchronize()
    model = lane_model.initial_state(params_1k, n, pos, vel, ypr, as_reset_kernel=True)
    got = env.state.cpu().numpy()
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32))
    mask = rng.random(n) < 0.3
    env.reset(mask=mask)                                   # params' own pose for the masked drones only
    torch.cuda.synchronize()
    again = env.state.cpu().numpy()
    assert np.array_equal(again[:, :n][:, ~mask], got[:, :n][:, ~mask])
    fresh = lane_model.initial_state(params_1k, n)
    assert np.array_equal(again[:, :n][:, mask], fresh[:, :n][:, mask])


def test_step_return_triple_matches_reference(params_1k):
    """Drone.step -> (R.T, E(rates as radians), R_new @ acc), components.py:247-248."""
    g = load_golden("g3_ema_noise")
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(params_1k, n)
    env.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    a = torch.from_numpy(acts).to(DEV)
    for t in range(T):
        out = env.step(a[t], wind_velocity_vector=np.zeros(3), object_list=[])
    RT, gyro, acc = (x.cpu().numpy() for x in out)
    np.testing.assert_allclose(RT, g["ret_RT"], atol=1e-5)
    np.testing.assert_allclose(gyro, g["ret_gyro"], atol=2e-4)     # rates ~ tens of deg used as radians
    np.testing.assert_allclose(acc, g["accel"][:, -1], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(env.position.cpu().numpy(), g["state"][:, -1, 0:3], rtol=1e-5, atol=1e-5)
    assert env.done.dtype == torch.bool and not env.done.any()


def test_return_triple_kernel_equals_host_build_and_torch_form(params_1k):
    """fpv_return_triple (what DroneBatch.step returns for fp32 state): rotation_matrix.T and E(rates) for 777 drones after
    300 steps of noise sticks, bit for bit the host build of the same instructions, within 2e-6 / 2e-4 of the float64
    formula (rates of tens of deg/s used as radians), the accel rows copied through; one launch instead of ~30."""
    n = 777
    env = _drone_batch(params_1k, n)
    env.reset()
    a = torch.from_numpy(sticks.ema_noise(300, range(n), seed=9)).to(DEV)
    env.rollout(a[:299])
    RT, gyro, acc = env.step(a[299])
    torch.cuda.synchronize()
    st = env.state.cpu().numpy()
    rt_h, gy_h = lane_model.return_matrices(st, n)
    assert np.array_equal(RT.cpu().numpy().view(np.uint32), rt_h.view(np.uint32))
    assert np.array_equal(gyro.cpu().numpy().view(np.uint32), gy_h.view(np.uint32))
    assert torch.equal(acc, env.accel[:, :n].t())
    q = st[6:10, :n].T.astype(np.float64)
    np.testing.assert_allclose(RT.cpu().numpy(), np.transpose(oracle.quat_to_matrix(q), (0, 2, 1)), atol=2e-6)
    r = st[10:13, :n].T.astype(np.float64)
    cr, sr, cp, sp, cy, sy = np.cos(r[:, 0]), np.sin(r[:, 0]), np.cos(r[:, 1]), np.sin(r[:, 1]), np.cos(r[:, 2]), np.sin(r[:, 2])
    E = np.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                  -sp, cp * sr, cp * cr], axis=1).reshape(n, 3, 3)
    np.testing.assert_allclose(gyro.cpu().numpy(), E, atol=2e-4)
    assert np.abs(r).max() > 20, "the scenario must reach rates of tens of deg/s"
    # fp16 handles keep the tensor-op form; a batch built without accel rows returns None for the third member
    lean = _drone_batch(params_1k, 8, with_accel=False)
    lean.reset()
    out = lean.step(a[0, :8].contiguous())
    assert out[2] is None and out[0].shape == (8, 3, 3)


def test_fp16_state_widened_by_one_kernel_equals_the_host_decoder(params_1k):
    """fpv_widen_state (what rows_f32 / position / velocity / quaternion / FpvVecEnv.obs read for fp16 storage): the eleven
    16-bit words of every drone decoded exactly as the step kernel decodes them - bit for bit the host build of
    fpv_unpack_half on the same storage words (v with its 5-bit low words, q rebuilt from its three stored components),
    position rows copied - at a ragged size, after a flight."""
    from fpyv_amd.env import FpvVecEnv
    n = 4099
    env = _drone_batch(params_1k, n, fp16_state=True, with_accel=False)
    env.reset()
    a = torch.from_numpy(sticks.ema_noise(50, range(n), seed=4)).to(DEV)
    env.rollout(a)
    got = env.rows_f32(0, 14)                                   # [n, 14]
    torch.cuda.synchronize()
    want = lane_model.join_half(env.state.cpu().numpy(), env.state_h.cpu().numpy().view(np.uint16))[:, :n].T
    assert got.shape == (n, 14) and np.array_equal(got.cpu().numpy().view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    wt = torch.from_numpy(np.ascontiguousarray(want)).to(DEV)
    assert torch.equal(env.quaternion, wt[:, 6:10]) and torch.equal(env.position, wt[:, 0:3]) and torch.equal(env.prev_thrust, wt[:, 13])
    assert float((env.quaternion.norm(dim=1) - 1).abs().max()) < 1e-6, "a stored attitude decodes to a unit quaternion"
    words = env.storage_words()
    assert words.shape == (11, env.ld) and words.dtype == torch.int16
    assert torch.equal(words[0, :n].view(torch.float16).float(), (got[:, 3].view(torch.int32) & ~0x1fff).view(torch.float32)), "vx: its binary16 part is the top of the decoded value"
    ve = FpvVecEnv(params_1k, num_envs=64, device=DEV, fp16_state=True)
    o0 = ve.reset()
    o1, r, d, info = ve.step(a[0, :64].contiguous())
    assert o1.shape == (64, 13) and o1.data_ptr() != o0.data_ptr() and bool(torch.isfinite(o1).all())
    torch.cuda.synchronize()
    w64 = lane_model.join_half(ve.batch.state.cpu().numpy(), ve.batch.state_h.cpu().numpy().view(np.uint16))[:13, :64].T
    assert np.array_equal(o1.cpu().numpy(), w64)


def test_default_fps60(params_60):
    g = load_golden("g1b_fps60_sin")
    n = g["actions"].shape[1]
    got = _run_golden(params_60, g).state.cpu().numpy()
    ref = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    oracle.drone_run(params_60, ref, g["actions"].astype(np.float64))
    assert_parity(soa_vs_oracle(got, ref, n), REL_TOL, "fps60")


def test_config1_first_1000_steps_and_10k_drift(params_1k):
    g = load_golden("g1_zero_10k")
    env = _drone_batch(params_1k, 1)
    env.reset()
    a = torch.zeros((1000, 1, 4), dtype=torch.float32, device=DEV)
    env.rollout(a)
    ref = oracle.drone_initial_state(1, [0, 0, 10.0], [1.0, 0, 0], [0, 0, 0])
    oracle.drone_run(params_1k, ref, np.zeros((1, 4)), steps=1000)
    assert_parity(soa_vs_oracle(env.state.cpu().numpy(), ref, 1), REL_TOL, "config 1 @1000")
    for _ in range(9):
        env.rollout(a)
    torch.cuda.synchronize()
    end = env.state.cpu().numpy()[:, 0]
    # reference end state: p=[2.5945630, 0, 206.3184240]; plain fp32 accumulation over
    # 10 000 steps holds ~1e-4 (0.02 m increments against 200 m), documented in DESIGN.md
    np.testing.assert_allclose(end[0:3], g["state"][0, -1, 0:3], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(end[6:10], [1, 0, 0, 0], atol=1e-6)


@pytest.mark.parametrize("n", [1, 63, 64, 127, 129, 257, 1000, 4096 + 5])
def test_bitwise_equal_to_lane_model_ragged_sizes(params_1k, n):
    """Empty tails, ragged sizes around the wave (64) and workgroup (128) widths: the gfx950 kernel must reproduce
    the host build of the same arithmetic bit for bit (both -ffp-contract=off, explicit fmaf)."""
    steps = 50
    acts = sticks.ema_noise(steps, range(n), seed=11)
    acts[:, :, 3] += np.float32(0.1)
    env = _drone_batch(params_1k, n, with_done_bits=True)
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV), wind=(1.0, -2.0, 0.5))
    torch.cuda.synchronize()
    got = env.state.cpu().numpy()
    model = lane_model.initial_state(params_1k, n)
    _, acc, done, rew = lane_model.run(params_1k, model, acts, wind=(1.0, -2.0, 0.5))
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32))
    assert np.array_equal(env.reward.cpu().numpy().view(np.uint32), rew.view(np.uint32))
    assert np.array_equal(env.done_u8.cpu().numpy(), done)
    assert np.array_equal(env.accel.cpu().numpy()[:, :n].view(np.uint32), acc[:, :n].view(np.uint32))
    assert np.all(got[:, n:] == 0), "padding columns beyond n must stay untouched"


def test_edge_inputs_bitwise_equal_to_lane_model(params_1k):
    """The two places where the kernel does NOT execute the host's instructions - the clip (one v_med3_f32 against
    fminf(fmaxf())) and the square root (v_sqrt_f32 + correction, arguments below the smallest normal flushed to zero,
    against sqrtf) - on the inputs a flight never produces: NaN and infinite sticks, sticks far outside [-1, 1],
    velocities whose square is subnormal, sits on the flush boundary, or is near the top of the fp32 range."""
    vels = np.array([[0, 0, 0], [1e-30, 0, 0], [1e-20, -1e-20, 1e-21], [1.05e-19, 0, 0], [1.1e-19, 0, 0], [7.7e-20, 7.7e-20, 0],
                     [1e-15, 0, 0], [3e18, -2e18, 1e18], [1e19, 0, 0], [-0.0, 0.0, -0.0]], dtype=np.float32)
    sticks_ = np.array([[np.nan, 0, 0, 0], [0, np.nan, np.nan, -0.5], [np.inf, -np.inf, 1e30, 0.2], [5, -7, 1.0000001, -1.5],
                        [-1, 1, -1, 1], [0, 0, 0, 0]], dtype=np.float32)
    n = len(vels) * len(sticks_)
    vel = np.repeat(vels, len(sticks_), axis=0)
    act = np.tile(sticks_, (len(vels), 1))[None].repeat(3, axis=0).copy()          # three steps of the same sticks
    for fused in (False, True):
        env = _drone_batch(params_1k, n)
        env.reset(velocity=vel)
        a = torch.from_numpy(act).to(DEV)
        if fused:
            env.rollout(a)
        else:
            for t in range(3):
                env.step(a[t], wind_velocity_vector=(0.0, 0.0, 0.0), return_imu=False)
        torch.cuda.synchronize()
        model = lane_model.initial_state(params_1k, n, velocity=vel)
        _, acc, done, rew = lane_model.run(params_1k, model, act)
        got = env.state.cpu().numpy()[:, :n]
        both_nan = np.isnan(got) & np.isnan(model[:, :n])                          # a NaN's payload is not part of the contract
        same = (got.view(np.uint32) == model[:, :n].view(np.uint32)) | both_nan
        assert same.all(), (fused, np.argwhere(~same)[:5], got[~same][:5], model[:, :n][~same][:5])
        # NaN rate sticks come out of the clip as -max_rates on both sides (fminf(fmaxf(NaN, lo), hi) = lo = v_med3's answer)
        np.testing.assert_allclose(got[10, 0::len(sticks_)], -0.7 * params_1k.max_rates * (1 + 0.3 + 0.09), rtol=1e-6)
        assert np.array_equal(env.done_u8.cpu().numpy(), done)


def test_config2_4096_drones_vs_oracle(params_1k):
    """BASELINE config 2: 4096 drones, constant throttle + sinusoidal roll/pitch, 1000 steps."""
    n, T = 4096, 1000
    acts = sticks.sinusoid(T, n, params_1k.dt)
    env = _drone_batch(params_1k, n)
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV))
    torch.cuda.synchronize()
    ref = oracle.drone_initial_state(n, params_1k.init_position, params_1k.init_velocity, [0, 0, 0])
    oracle.drone_run(params_1k, ref, acts.astype(np.float64), threads=0)
    err = soa_vs_oracle(env.state.cpu().numpy(), ref, n)
    assert_parity(err, REL_TOL, "config 2")
    # spot value of SURVEY App. B (drone 0 == phase 0)
    np.testing.assert_allclose(env.position[0].cpu().numpy(),
                               [1.4918055996320558, -1.6771575108044852, 10.247650148078797], rtol=1e-5)


def test_ground_contact_done_sequence(params_1k):
    g = load_golden("g6_ground")
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(params_1k, n)
    env.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    dones = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
    rewards = torch.zeros((T, n), dtype=torch.float32, device=DEV)
    env.rollout(torch.from_numpy(acts).to(DEV), rewards=rewards, dones=dones)
    torch.cuda.synchronize()
    seq = dones.cpu().numpy().T
    assert np.array_equal(seq, g["done"]), "done must flip on exactly the reference's steps"
    assert seq[2].any() and not seq[2][-1], "done is recomputed every step, not latched (components.py:236)"


def test_ground_plane_contact_vs_reference_capture(params_1k):
    """FPV_FLAG_GROUND = Drone.step(..., object_list=[Ground]) (components.py:198-214)."""
    g = load_golden("g9_ground_contact")
    p = params_1k.replace(ground=True)
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(p, n)
    model = lane_model.initial_state(p, n, g["init_position"], g["init_velocity"], g["init_ypr"])
    env.state[:, :n] = torch.from_numpy(model[:, :n]).to(DEV)     # identical fp32 start for the bitwise check
    dones = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
    env.rollout(torch.from_numpy(acts).to(DEV), dones=dones)
    torch.cuda.synchronize()
    assert np.array_equal(dones.cpu().numpy().T, g["done"])
    ref = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1],
                          g["prev_thrust"][:, -1:]], axis=1)
    got = env.state.cpu().numpy()
    err = soa_vs_oracle(got, ref, n)
    assert err["pos_comp"] < REL_TOL and err["quat_abs"] < REL_TOL, err      # measured 1.2e-6 (tests/test_lane_model.py)
    lane_model.run(p, model, acts)
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32))


def test_reset_mask_and_per_drone_initial_conditions(params_1k):
    n = 300
    rng = np.random.default_rng(3)
    pos = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    vel = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    ypr = rng.uniform(-170, 170, (n, 3)).astype(np.float32)
    env = _drone_batch(params_1k, n)
    env.reset(position=pos, velocity=vel, ypr=ypr)
    torch.cuda.synchronize()
    s = env.state.cpu().numpy()
    assert np.array_equal(s[0:3, :n].T, pos) and np.array_equal(s[3:6, :n].T, vel)
    want = lane_model.initial_state(params_1k, n, pos, vel, ypr)
    np.testing.assert_allclose(s[6:10, :n], want[6:10, :n], atol=3e-7)
    assert np.all(s[10:14] == 0)
    # step a bit, then reset only the even drones to the defaults
    env.rollout(torch.from_numpy(sticks.ema_noise(20, range(n), seed=1)).to(DEV))
    before = env.state.clone()
    mask = np.zeros(n, dtype=np.uint8)
    mask[::2] = 1
    env.reset(mask=mask)
    torch.cuda.synchronize()
    after = env.state.cpu().numpy()
    assert np.array_equal(after[:, 1:n:2], before.cpu().numpy()[:, 1:n:2])
    np.testing.assert_array_equal(after[0:3, 0:n:2].T, np.broadcast_to(params_1k.init_position.astype(np.float32), (n // 2, 3)))
    assert np.all(after[10:14, 0:n:2] == 0)


def test_auto_reset_ceiling_and_episode_stats(params_1k):
    p = params_1k.replace(ceiling=0.8, init_position=np.array([0.0, 0.0, 0.3]))   # ground 0.25 s below, ceiling above
    n = 512
    env = _drone_batch(p, n, auto_reset=True, track_episodes=True, with_done_bits=True)
    env.reset()
    acts = np.zeros((n, 4), dtype=np.float32)
    acts[:, 3] = np.linspace(-1, 1, n)
    a = torch.from_numpy(acts).to(DEV)
    model = lane_model.initial_state(p, n)
    ep_ret = np.zeros(n, dtype=np.float32)
    ep_len = np.zeros(n, dtype=np.int32)
    finished = np.zeros(n, dtype=np.int64)
    for t in range(400):
        env.step(a, return_imu=False)
        _, _, done, rew = lane_model.run(p, model, acts, steps=1, auto_reset=True)
        ep_ret += rew
        ep_len += 1
        d = done.astype(bool)
        torch.cuda.synchronize()
        assert np.array_equal(env.done_u8.cpu().numpy(), done)
        bits = env.done_bits.cpu().numpy().view(np.uint64)
        unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(-1)[:n].astype(np.uint8)
        assert np.array_equal(unpacked, done), "wave-ballot bit mask != byte mask"
        if d.any():
            np.testing.assert_allclose(env.last_return.cpu().numpy()[d], ep_ret[d], rtol=1e-5)
            assert np.array_equal(env.last_length.cpu().numpy()[d], ep_len[d])
            finished[d] += 1
            ep_ret[d] = 0
            ep_len[d] = 0
        assert np.array_equal(env.ep_length.cpu().numpy(), ep_len)
    assert np.array_equal(env.state.cpu().numpy()[:, :n].view(np.uint32), model[:, :n].view(np.uint32))
    assert finished[-1] >= 1 and finished[0] >= 1, "both the ceiling and the ground must end episodes"
    assert np.all(np.abs(env.state[2, :n].cpu().numpy()) <= 0.9)


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


def test_broadcast_action_and_numpy_action(params_1k):
    env = _drone_batch(params_1k, 100)
    env.reset()
    env.step(np.array([0.5, 0, 0, 0]), return_imu=False)          # simulator.py:89 style single action
    env2 = _drone_batch(params_1k, 100)
    env2.reset()
    env2.step(torch.tensor([[0.5, 0, 0, 0]] * 100, device=DEV), return_imu=False)
    assert torch.equal(env.state, env2.state)


def _racer_replay(p, g):
    """Replay a Racer golden through RacerBatch, comparing at every snapshot; returns the worst errors."""
    from fpyv_amd.env import RacerBatch
    env = RacerBatch(p, 1, device=DEV)
    env.reset()
    acts = torch.from_numpy(g["actions"]).to(DEV)
    prev, worst = 0, dict(quat=0.0, pos=0.0, omega=0.0)
    for k, t in enumerate(np.asarray(g["snap_steps"]).reshape(-1)):
        env.rollout(acts[prev:int(t)].contiguous())
        prev = int(t)
        s = env.state.cpu().numpy()
        q = s[6:10, 0].astype(np.float64)
        x, y, z, w = g["quat_xyzw"][0, k]
        qr = np.array([w, x, y, z])
        q *= np.sign(q @ qr)
        pr = g["position"][0, k]
        worst["quat"] = max(worst["quat"], np.abs(q - qr).max())
        worst["pos"] = max(worst["pos"], np.abs(s[0:3, 0] - pr).max() / max(np.abs(pr).max(), 1e-3))
        worst["omega"] = max(worst["omega"], np.abs(s[10:13, 0].astype(np.float64) + s[20:23, 0] - g["omega"][0, k]).max())
    return env, worst


@pytest.mark.parametrize("name", ["g7_racer_main", "g8_racer_pid_thrust", "g15_racer_prop7"])
def test_racer_vs_reference_capture(params_1k, name):
    """Racer.step AS WRITTEN (omega radians per step) against the reference captures at the north-star bar,
    1e-5 on position and quaternion, at every snapshot of the 1000 steps: the rate loop and the attitude
    increment run in float64 with (hi, lo) state rows (229 B per env-step)."""
    from conftest import racer_params_for_golden
    g = load_golden(name)
    p = racer_params_for_golden(g)
    env, worst = _racer_replay(p, g)
    assert env.algorithmic_bytes() == 229 and env.state.shape[0] == 29
    assert worst["quat"] < REL_TOL and worst["pos"] < REL_TOL and worst["omega"] < 1e-8, worst
    model = lane_model.initial_state(p, 1)
    lane_model.run(p, model, g["actions"])
    assert np.array_equal(env.state.cpu().numpy()[:, :1].view(np.uint32), model[:, :1].view(np.uint32)), \
        "kernel != lane model (bitwise; the float64 rate loop is library-free: same fma sequence on both sides)"


def test_racer_with_components_pid_vs_reference_capture(params_1k):
    """a16 as the Racer's rate loop (racer_pid_variant = 1) against capture G12."""
    from test_oracle_golden import _cpid_params
    g = load_golden("g12_racer_components_pid")
    p = _cpid_params(params_1k, g)
    env, worst = _racer_replay(p, g)
    assert env.algorithmic_bytes() == 229 + 24
    assert worst["quat"] < REL_TOL and worst["pos"] < REL_TOL, worst
    s = env.state.cpu().numpy()
    np.testing.assert_allclose(s[26:29, 0], g["prev_derivative"][0, -1], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(s[13:16, 0].astype(np.float64) + s[23:26, 0], g["i_error"][0, -1], rtol=1e-6, atol=1e-9)


def test_components_pid_kernel_vs_reference_class():
    """a16 standalone: fpyv_amd.components.PID (N controllers per launch) against the reference class's
    outputs on the seeded sequences of G11, all four gain sets side by side in one batch of 4 x 64
    controllers; bitwise against the host build of the same fp32 arithmetic."""
    from fpyv_amd.components import PID
    g = load_golden("g11_components_pid")
    T = g["current"].shape[1]
    for c in range(g["gains"].shape[0]):
        kP, kI, kD, dt, ic, lo, hi, dtr = g["gains"][c]
        pid = PID(kP, kI, kD, dt, integral_clip=ic, min_output=lo, max_output=hi, derivative_transition_rate=dtr,
                  num_envs=64, device=DEV)
        cur = torch.from_numpy(g["current"][c]).float().to(DEV)
        tgt = torch.from_numpy(g["target"][c]).float().to(DEV)
        outs = torch.zeros((T, 64), device=DEV)
        for t in range(T):
            outs[t] = pid(cur[t].expand(64), tgt[t].expand(64) if c % 2 else float(tgt[t]))
        torch.cuda.synchronize()
        o = outs.cpu().numpy()
        assert np.all(o == o[:, :1]), "every lane runs the same controller"
        np.testing.assert_allclose(o[:, 0], g["out"][c], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(pid.integral.cpu().numpy(), g["integral"][c][-1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(pid.error.cpu().numpy(), g["error"][c][-1], rtol=1e-6, atol=1e-7)
        want, st = lane_model.pid_run(g["gains"][c], g["current"][c], g["target"][c])
        assert np.array_equal(o[:, 0].view(np.uint32), want.view(np.uint32)), "kernel != lane model (bitwise)"
        pid.reset(mask=torch.arange(64, device=DEV) % 2 == 0)
        torch.cuda.synchronize()
        assert float(pid.integral[0]) == 0.0 and float(pid.state[3, 0]) == 1.0 and float(pid.state[3, 1]) == 0.0


def test_racer_omega_dt_batch_vs_oracle(params_1k):
    from fpyv_amd.env import RacerBatch
    n, T = 500, 400
    rng = np.random.default_rng(9)
    acts = np.concatenate([rng.uniform(-6, 6, (T, n, 3)), rng.uniform(0, 8, (T, n, 1))], axis=2).astype(np.float32)
    p = params_1k.replace(mode=1, racer_omega_dt=True, racer_pid=np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0]]))
    env = RacerBatch(p, n, device=DEV)
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV))
    torch.cuda.synchronize()
    s = env.state.cpu().numpy()
    ref = oracle.racer_initial_state(n)
    oracle.racer_run(p, ref, acts.astype(np.float64), threads=0)
    np.testing.assert_allclose(s[0:3, :n].T, ref[:, 0:3], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(s[10:13, :n].T, ref[:, 10:13], rtol=1e-5, atol=1e-5)
    q = s[6:10, :n].T.astype(np.float64)
    qr = ref[:, [9, 6, 7, 8]]
    q *= np.sign(np.sum(q * qr, axis=1, keepdims=True))
    assert np.abs(q - qr).max() < 1e-5


def test_full_size_properties_1M(params_1k):
    """BASELINE config 3 size (2^20 drones): properties that need no full-size oracle run.
    N-invariance (a drone's trajectory does not depend on the batch it sits in or on its lane),
    unit quaternions, done bit mask == byte mask, every host path (k-step kernel, k launches, hipGraph replay) agrees bitwise."""
    n, k = 1 << 20, 64
    p = params_1k
    acts = sticks.ema_noise_device(k, n, DEV, seed=1234)
    states = []
    for j, how in enumerate((dict(fused=True), dict(fused=False), dict(graph=True))):
        env = _drone_batch(p, n, with_accel=False, with_done_bits=True)
        env.reset()
        env.rollout(acts, **how)
        torch.cuda.synchronize()
        states.append(env.state.clone())
        if j != 2:
            del env
    assert torch.equal(states[0], states[1]) and torch.equal(states[0], states[2])
    s = states[0]
    qn = torch.linalg.vector_norm(s[6:10, :n], dim=0)
    assert float((qn - 1).abs().max()) < 5e-7
    assert bool(torch.isfinite(s).all())
    # sample 2048 drones spread over the batch (first/last lanes, block and wave edges) and replay
    # them in a small batch and on the host lane model
    idx = np.unique(np.concatenate([np.arange(0, 256), np.arange(n - 256, n),
                                    np.random.default_rng(0).integers(0, n, 1536)]))
    sub = acts[:, torch.from_numpy(idx).to(DEV)].contiguous()
    small = _drone_batch(p, len(idx), with_accel=False)
    small.reset()
    small.rollout(sub)
    torch.cuda.synchronize()
    assert torch.equal(small.state[:, :len(idx)], s[:, torch.from_numpy(idx).to(DEV)])
    model = lane_model.initial_state(p, len(idx))
    lane_model.run(p, model, sub.cpu().numpy())
    assert np.array_equal(model[:, :len(idx)].view(np.uint32), small.state[:, :len(idx)].cpu().numpy().view(np.uint32))
    ref = oracle.drone_initial_state(len(idx), p.init_position, p.init_velocity, [0, 0, 0])
    oracle.drone_run(p, ref, sub.cpu().numpy().astype(np.float64), threads=0)
    assert_parity(soa_vs_oracle(model, ref, len(idx)), REL_TOL, "1M sample")
    bits = env.done_bits.cpu().numpy().view(np.uint64)
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(-1)[:n]
    assert np.array_equal(unpacked.astype(np.uint8), env.done_u8.cpu().numpy())


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


# ---- BASELINE config 4: fp16 state / fp32 integrator -------------------------------------------------
def test_fp16_state_bitwise_vs_lane_model_and_restated_tolerance(params_1k):
    from test_lane_model import FP16_TOL
    g = load_golden("g3_ema_noise")
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(params_1k, n, fp16_state=True, rounding_seed=5, with_accel=False)
    assert env.algorithmic_bytes() == 89 and env.state.shape[0] == 3 and env.state_h.dtype == torch.float16 and env.state_h.numel() == 11 * env.ld
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV))
    torch.cuda.synchronize()
    pos, sh = lane_model.split_half(lane_model.initial_state(params_1k, n), seed=5)
    lane_model.run_h(params_1k, pos, sh, acts, seed0=5)
    assert np.array_equal(env.state.cpu().numpy()[:, :n].view(np.uint32), pos[:, :n].view(np.uint32))
    ld, lm = env.ld, pos.shape[1]              # the batch pads its row stride, the lane model does not
    got_h, want_h = env.state_h.cpu().numpy().view(np.uint16), sh
    assert np.array_equal(got_h[:10 * ld].reshape(5, ld, 2)[:, :n], want_h[:10 * lm].reshape(5, lm, 2)[:, :n])   # pair rows
    assert np.array_equal(got_h[10 * ld:10 * ld + n], want_h[10 * lm:10 * lm + n])                              # thrust halves
    ref = oracle.drone_initial_state(n, params_1k.init_position, params_1k.init_velocity, [0, 0, 0])
    oracle.drone_run(params_1k, ref, acts.astype(np.float64))
    got = lane_model.join_half(env.state.cpu().numpy(), got_h)
    err = soa_vs_oracle(got, ref, n)
    for k, tol in FP16_TOL.items():
        assert err[k] <= tol, (k, err[k])
    np.testing.assert_allclose(env.velocity.cpu().numpy(), ref[:, 3:6], rtol=3e-2, atol=4e-2)      # per component (attitude error x thrust); the norm-based bound is vel_rel above


@pytest.mark.parametrize("n", [1, 63, 333, 4099])
@pytest.mark.parametrize("fused", [False, True], ids=["single-step", "k-step"])
def test_fp16_state_ragged_sizes_vs_lane_model_and_oracle(params_1k, n, fused):
    """the fp16 kernels' odd-n path (st_thrust_pair_h: the last even lane has no live neighbour and the DPP
    quad-permute hands it a zero half) against an INDEPENDENT restatement - the host lane model, bit for bit, and the
    float64 oracle within the restated tolerance - for n = 1 (one lane), 63 (odd, inside one wave), 333 (odd, last
    workgroup partly filled) and 4099 (odd, 33 workgroups)."""
    from test_lane_model import FP16_TOL
    steps = 1000 if n <= 333 else 250
    acts = sticks.ema_noise(steps, range(n), seed=31)
    env = _drone_batch(params_1k, n, fp16_state=True, rounding_seed=17, with_accel=False)
    env.reset()
    a = torch.from_numpy(acts).to(DEV)
    if fused:
        for t0 in range(0, steps, 125):
            env.rollout(a[t0:t0 + 125])                    # fpv_step_n: fpv_drone_rollout_h_kernel
    else:
        for t in range(steps):
            env.step(a[t], return_imu=False)               # fpv_step: fpv_drone_step_h_kernel
    torch.cuda.synchronize()
    pos, sh = lane_model.split_half(lane_model.initial_state(params_1k, n), seed=17)
    done, rew = lane_model.run_h(params_1k, pos, sh, acts, seed0=17)
    ld, lm = env.ld, pos.shape[1]
    got_h = env.state_h.cpu().numpy().view(np.uint16)
    assert np.array_equal(env.state.cpu().numpy()[:, :n].view(np.uint32), pos[:, :n].view(np.uint32))
    assert np.array_equal(got_h[:10 * ld].reshape(5, ld, 2)[:, :n], sh[:10 * lm].reshape(5, lm, 2)[:, :n])      # pair rows
    assert np.array_equal(got_h[10 * ld:10 * ld + n], sh[10 * lm:10 * lm + n]), "thrust halves (the exchanged row)"
    assert np.array_equal(env.done_u8.cpu().numpy(), done) and np.array_equal(env.reward.cpu().numpy().view(np.uint32), rew.view(np.uint32))
    # nothing beyond the batch's padded pair of the last drone is written: halves n+1.. of the thrust row stay zero
    assert not got_h[10 * ld + n + (n & 1):11 * ld].any() and not got_h[:10 * ld].reshape(5, ld, 2)[:, n:].any()
    ref = oracle.drone_initial_state(n, params_1k.init_position, params_1k.init_velocity, [0, 0, 0])
    oracle.drone_run(params_1k, ref, acts.astype(np.float64), threads=0)
    err = soa_vs_oracle(lane_model.join_half(env.state.cpu().numpy(), got_h), ref, n)
    for k, tol in FP16_TOL.items():
        assert err[k] <= tol, (k, err[k], n)


def test_fp16_device_conversions_equal_the_host_emulation_on_special_values(params_1k):
    """v_cvt_pkrtz_f16_f32 (round toward zero, two values per instruction) and the 13-bit stochastic rounding on the
    device against the host emulation the lane model uses, on the values a trajectory never visits: every exponent
    from fp32 subnormals to overflow, subnormal halves, the saturation boundary, +-0 (a non-finite state is garbage on
    either side and is not compared).  The values reach the
    kernel's packer as per-drone reset velocities (fpv_reset_kernel packs with the buffer's rounding seed)."""
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 65504, 65519.9, 65520, 65535.9, 65536, 7e4, -7e4, 3e38, -3e38, 6e-8, 5.97e-8, 5.9e-8, 3e-8, 1e-41,
                        6.1e-5, 6.09e-5, 6.103515625e-5, 1.0, -1.0, 1.0009765625, 1.00097, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25], dtype=np.float32)
    rnd = (rng.standard_normal(3 * 2000 - len(special)) * 10.0 ** rng.integers(-12, 7, 3 * 2000 - len(special))).astype(np.float32)
    vel = np.concatenate([special, rnd]).reshape(-1, 3)
    n = vel.shape[0]
    env = _drone_batch(params_1k, n, fp16_state=True, rounding_seed=4242, with_accel=False)
    env.reset(velocity=vel)
    torch.cuda.synchronize()
    ld = env.ld
    got = env.state_h.cpu().numpy().view(np.uint16)[:10 * ld].reshape(5, ld, 2)[:, :n]        # [pair row, drone, half]
    for i in range(n):
        st = np.zeros(14, dtype=np.float32)
        st[0:3] = [0, 0, 10]; st[3:6] = vel[i]; st[6] = 1.0
        w = lane_model.pack_state(st, 4242, i)
        want = [(int(w[0]) & 0xffff, int(w[0]) >> 16), (int(w[1]) & 0xffff, int(w[1]) >> 16)]
        assert (int(got[0, i, 0]), int(got[0, i, 1])) == want[0], (i, vel[i], got[0, i], [hex(x) for x in want[0]])
        assert (int(got[1, i, 0]), int(got[1, i, 1])) == want[1], (i, vel[i], got[1, i], [hex(x) for x in want[1]])


@pytest.mark.parametrize("fused", [False, True], ids=["single-step", "k-step"])
def test_fp16_state_is_shard_invariant(params_1k, fused):
    """The stochastic rounding of a drone is keyed by its GLOBAL id (drone_id_offset + lane), like its stick-noise stream:
    the fp16 trajectory of a drone must not depend on the shard it lands in or on its lane (round 2 keyed it by the
    local lane index).  One batch of 3000 drones against the same drones as shards of 1000 / 77 / 1923."""
    n, steps = 3000, 120
    acts = sticks.ema_noise(steps, range(n), seed=5)
    a = torch.from_numpy(acts).to(DEV)

    def run(lo, hi):
        env = _drone_batch(params_1k, hi - lo, fp16_state=True, rounding_seed=3, with_accel=False, drone_id_offset=lo)
        env.reset()
        sub = a[:, lo:hi].contiguous()
        if fused:
            env.rollout(sub)
        else:
            for t in range(steps):
                env.step(sub[t], return_imu=False)
        torch.cuda.synchronize()
        m = hi - lo
        half = env.state_h.cpu().numpy().view(np.uint16)
        ld = env.ld
        return env.state.cpu().numpy()[:, :m], half[:10 * ld].reshape(5, ld, 2)[:, :m], half[10 * ld:10 * ld + m]

    whole = run(0, n)
    for lo, hi in ((0, 1000), (1000, 1077), (1077, 3000)):
        part = run(lo, hi)
        assert np.array_equal(part[0].view(np.uint32), whole[0][:, lo:hi].view(np.uint32)), (lo, hi)
        assert np.array_equal(part[1], whole[1][:, lo:hi]) and np.array_equal(part[2], whole[2][lo:hi]), (lo, hi)
    pos, sh = lane_model.split_half(lane_model.initial_state(params_1k, 77), seed=3, drone_id_offset=1000)
    lane_model.run_h(params_1k, pos, sh, acts[:, 1000:1077], seed0=3, drone_id_offset=1000)
    assert np.array_equal(whole[0][:, 1000:1077].view(np.uint32), pos[:, :77].view(np.uint32)), "and the host build agrees on the keyed stream"


def test_fp16_state_full_size_vs_fp32_run():
    """Config 4 at full size: same sticks through the fp32 and the fp16-storage kernels; the
    distribution of the difference after 500 steps must sit inside the restated tolerance."""
    n, k = 1 << 20, 500
    p = load_params(fps=1000)
    acts = sticks.ema_noise_device(50, n, DEV, seed=99)
    e32 = _drone_batch(p, n, with_accel=False)
    e16 = _drone_batch(p, n, with_accel=False, fp16_state=True)
    e32.reset(); e16.reset()
    for _ in range(k // 50):
        e32.rollout(acts); e16.rollout(acts)
    torch.cuda.synchronize()
    dp = (e16.position - e32.position).norm(dim=1) / e32.position.norm(dim=1)
    q16, q32 = e16.quaternion, e32.quaternion
    dq = (q16 * torch.sign((q16 * q32).sum(dim=1, keepdim=True)) - q32).abs().amax(dim=1)
    # round 3 (eleven binary16 values): max 2e-2 / mean 2e-3 for p, max 3e-2 / mean 3e-3 for q were the asserted bounds; with
    # 15 mantissa bits for v and the smallest-three quaternion the same 22 bytes hold these, over 2^20 drones
    assert float(dp.max()) < 5e-3 and float(dp.mean()) < 5e-4, (float(dp.max()), float(dp.mean()))
    assert float(dq.max()) < 8e-3 and float(dq.mean()) < 8e-4, (float(dq.max()), float(dq.mean()))
    assert bool(torch.isfinite(e16.rows_f32(0, 14)).all())


@pytest.mark.parametrize("n", [1, 63, 1000, 4096 + 5, 1 << 16])
def test_obs_aos_rows_equal_soa_state(params_1k, n):
    """The LDS-transposed [n, 16] observation must be exactly the SoA state + the accelerometer
    triple of the same step (p3 v3 q4 rates3 R_new@acc 3), for ragged sizes."""
    steps = 20
    acts = torch.from_numpy(sticks.ema_noise(steps, range(min(n, 2048)), seed=2)).to(DEV)
    if n > 2048:
        acts = acts.repeat(1, (n + 2047) // 2048, 1)[:, :n].contiguous()
    env = _drone_batch(params_1k.replace(ceiling=10.4), n, with_obs_aos=True, auto_reset=True)
    plain = _drone_batch(params_1k.replace(ceiling=10.4), n, auto_reset=True)
    env.reset(); plain.reset()
    env.obs_aos.fill_(float("nan"))
    for t in range(steps):
        env.step(acts[t], return_imu=False)
        plain.step(acts[t], return_imu=False)
    torch.cuda.synchronize()
    assert torch.equal(env.state, plain.state), "the AoS head must not change the physics"
    obs = env.obs_aos
    assert obs.shape == (n, 16)
    assert torch.equal(obs[:, 0:13], env.state[0:13, :n].t())
    assert torch.equal(obs[:, 13:16], env.accel[:, :n].t())
    assert torch.equal(env.reward, plain.reward) and torch.equal(env.done_u8, plain.done_u8)


def test_object_list_collisions_vs_reference_capture(params_1k):
    """Drone.step(..., object_list=[Target, Cylinder, Cylinder, Ground]) with a moving target
    (simulator.py:85-87), through the public API, against the reference capture G10."""
    from fpyv_amd.objects import Cylinder, Ground, Target
    from test_oracle_golden import _g10_objects
    g = load_golden("g10_objects")
    acts = g["actions"]
    T, n = acts.shape[:2]
    target = Target([0.0, -6.0, 3.0], 0.8, path={"radius": 1.5, "resolution": 20000})
    objs = [target, Cylinder([3.0, 0.0, 0.0], 1.0, 5.0), Cylinder([-2.0, 2.5, 0.0], 0.6, 1.5), Ground()]
    env = _drone_batch(params_1k, n)
    model = lane_model.initial_state(params_1k, n, g["init_position"], g["init_velocity"], g["init_ypr"])
    env.state[:, :n] = torch.from_numpy(model[:, :n]).to(DEV)
    a = torch.from_numpy(acts).to(DEV)
    seq = np.zeros((n, T), dtype=np.uint8)
    try:
        for t in range(T):
            target.update()
            np.testing.assert_allclose(target.position, g["target_positions"][t], atol=1e-12)
            env.step(a[t], wind_velocity_vector=np.zeros(3), object_list=objs, return_imu=False)
            seq[:, t] = env.done_u8.cpu().numpy()
            lane_model.set_objects(_g10_objects(g, t))
            lane_model.run(params_1k, model, acts[t:t + 1])
    finally:
        lane_model.set_objects(())
    got = env.state.cpu().numpy()
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32)), "kernel != lane model (bitwise)"
    first = lambda d: int(np.argmax(d)) if d.any() else -1      # noqa: E731
    for i in range(n):
        assert first(seq[i]) == first(g["done"][i]), "crash on exactly the reference's step"
    ok = ~g["done"].any(axis=1)
    ref = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1],
                          g["prev_thrust"][:, -1:]], axis=1)
    err = soa_vs_oracle(np.ascontiguousarray(got[:, np.flatnonzero(ok)]), ref[ok], int(ok.sum()))
    assert err["pos_comp"] < REL_TOL and err["quat_abs"] < REL_TOL, err      # measured 4.1e-6 (tests/test_lane_model.py)
    # an object list and FPV_FLAG_GROUND are mutually exclusive; too many objects are rejected
    with pytest.raises(ValueError):
        env.step(a[0], object_list=[Ground()] * 9)


def test_raised_objects_vs_reference_capture(params_1k):
    """Capture G16 through the public API: object_list = [Ground, Cylinder (z = 1.2 .. 3.2), standing Target, small raised
    Cylinder] - the cylinder normal's relative-vs-absolute height test of the reference (components.py:718-720), the rim
    from below, the top, Ground first in the list.  Single steps and one k-step launch: crashes on the reference's
    steps, survivors within 1e-5 of the reference's numbers, kernel == host build bit for bit."""
    from fpyv_amd.objects import Cylinder, Ground, Target
    g = load_golden("g16_objects_raised")
    acts = g["actions"]
    T, n = acts.shape[:2]
    objs = [Ground(), Cylinder([3.0, 0.0, 1.2], 1.0, 2.0), Target([0.0, 4.0, 2.0], 0.8), Cylinder([-2.0, -2.0, 0.8], 0.5, 0.6)]
    from fpyv_amd.objects import to_rows
    assert np.allclose(np.asarray(to_rows(objs), dtype=float), g["objects"])
    model = lane_model.initial_state(params_1k, n, g["init_position"], g["init_velocity"], g["init_ypr"])
    start = torch.from_numpy(model[:, :n].copy()).to(DEV)
    env, fused = _drone_batch(params_1k, n), _drone_batch(params_1k, n)
    env.state[:, :n] = start
    fused.state[:, :n] = start
    a = torch.from_numpy(acts).to(DEV)
    seq = np.zeros((n, T), dtype=np.uint8)
    dones = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
    try:
        lane_model.set_objects(tuple(tuple(o) for o in g["objects"]))
        for t in range(T):
            env.step(a[t], wind_velocity_vector=np.zeros(3), object_list=objs, return_imu=False)
            seq[:, t] = env.done_u8.cpu().numpy()
        lane_model.run(params_1k, model, acts)
    finally:
        lane_model.set_objects(())
    fused.rollout(a, dones=dones, object_list=objs)          # the same 800 steps in ONE launch
    torch.cuda.synchronize()
    got = env.state.cpu().numpy()
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32)), "kernel != lane model (bitwise)"
    assert torch.equal(fused.state, env.state) and np.array_equal(dones.cpu().numpy().T, seq)
    first = lambda d: int(np.argmax(d)) if d.any() else -1      # noqa: E731
    for i in range(n):
        assert first(seq[i]) == first(g["done"][i]), "crash on exactly the reference's step"
    ok = ~g["done"].any(axis=1)
    ref = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1],
                          g["prev_thrust"][:, -1:]], axis=1)
    err = soa_vs_oracle(np.ascontiguousarray(got[:, np.flatnonzero(ok)]), ref[ok], int(ok.sum()))
    assert ok.sum() == 3 and err["pos_comp"] < REL_TOL and err["quat_abs"] < REL_TOL, err


def test_config5_shard_invariance_8M(params_1k):
    """BASELINE config 5 at its full size on ONE GPU: 8 388 608 drones as one batch vs the same
    drones as 8 contiguous shards (what 8 ranks would own), in-kernel stick noise keyed by global
    drone id, auto-reset on ground contact or |z| > ceiling.  Every shard must reproduce its slice
    of the global batch bit for bit, and the concatenated done masks must equal the global mask."""
    from fpyv_amd.dist import shard_range, unpack_done_bits
    from fpyv_amd.env import DroneBatch
    n_total, world, steps = 1 << 23, 8, 24
    p = params_1k.replace(ceiling=10.02, noise_gain=3.0)      # tight ceiling + strong sticks: resets happen early
    kw = dict(device=DEV, auto_reset=True, stick_noise=True, noise_seed=2024, with_accel=False,
              with_done_bits=True)
    whole = DroneBatch(p, n_total, **kw)
    whole.reset()
    whole.rollout(None, steps=steps)
    torch.cuda.synchronize()
    assert int(whole.done_u8.sum()) > 0, "the scenario must end episodes in the last step too"
    g_state, g_done, g_bits = whole.state[:, :n_total].clone(), whole.done_u8.clone(), whole.done_bits.clone()
    any_reset = bool((g_state[13] == 0).any())               # freshly reset lanes have prev_thrust == 0
    del whole
    masks = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        sh = DroneBatch(p, hi - lo, drone_id_offset=lo, **kw)
        sh.reset()
        sh.rollout(None, steps=steps)
        torch.cuda.synchronize()
        assert torch.equal(sh.state[:, :hi - lo], g_state[:, lo:hi]), f"shard {r} differs from its slice"
        assert torch.equal(sh.done_u8, g_done[lo:hi])
        masks.append(sh.done_bits.clone())
        del sh
    assert torch.equal(torch.cat(masks), g_bits), "all-gather of shard masks == global mask"
    assert torch.equal(unpack_done_bits(g_bits, n_total), g_done)
    assert any_reset, "the scenario must trigger in-kernel resets"


def test_maximum_handle_size_2_to_the_28(params_1k):
    """The largest population one handle takes (fpv_create: n <= 2^28, the bound that keeps 16 * i - the byte offset
    of a lane's action row - inside 32 bits): 268 435 456 drones, 15 GB of state + 4.3 GB of sticks on one GPU, three
    single steps and a 2-step k-step launch.  The first, the middle and the LAST 4096 drones (lane offsets up to
    0xFFFFFFF0) must equal the same drones stepped as small batches bit for bit; n + 1 is refused."""
    from fpyv_amd import _lib as L
    from fpyv_amd.env import DroneBatch
    n = 1 << 28
    free, _ = torch.cuda.mem_get_info()
    if free < 30 << 30:
        pytest.skip("needs 30 GB of free device memory")
    with pytest.raises(L.FpvError, match="2\\^28"):
        DroneBatch(params_1k, n + 1, device=DEV)

    def sticks_of(ids):                         # a per-drone stick pattern any slice can re-create from its ids alone
        x = ids.to(torch.float32) * 1e-3
        return torch.stack([torch.sin(x) * 0.6, torch.cos(x * 0.7) * 0.6, torch.sin(x * 1.3) * 0.3,
                            torch.cos(x * 0.31) * 0.5 - 0.2], dim=1).contiguous()

    big = DroneBatch(params_1k, n, device=DEV, with_accel=False)
    big.reset()
    a = torch.empty((n, 4), dtype=torch.float32, device=DEV)
    chunk = 1 << 24
    for lo in range(0, n, chunk):               # built in pieces: the temporaries stay small
        a[lo:lo + chunk] = sticks_of(torch.arange(lo, lo + chunk, device=DEV))
    for _ in range(3):
        big.step(a, return_imu=False)
    big.rollout(a, steps=2)
    torch.cuda.synchronize()
    for lo in (0, (n >> 1) - 2048, n - 4096):
        ids = torch.arange(lo, lo + 4096, device=DEV)
        small = DroneBatch(params_1k, 4096, device=DEV, with_accel=False)
        small.reset()
        sa = sticks_of(ids)
        assert torch.equal(sa, a[lo:lo + 4096])
        for _ in range(3):
            small.step(sa, return_imu=False)
        small.rollout(sa, steps=2)
        torch.cuda.synchronize()
        assert torch.equal(small.state[:, :4096], big.state[:, lo:lo + 4096]), f"drones {lo}.. differ"
        assert torch.equal(small.reward, big.reward[lo:lo + 4096]) and torch.equal(small.done, big.done[lo:lo + 4096])
    assert bool(torch.isfinite(big.state[:, :n]).all())
    del big, a
    torch.cuda.empty_cache()


def test_config1_10k_steps_with_kahan_rows(params_1k):
    """BASELINE config 1 end to end on the GPU at the 1e-5 bar: 10 000 zero-stick steps with the Kahan
    compensation rows; also bit-identical to the host lane model, and the rows reset with the lane."""
    g = load_golden("g1_zero_10k")
    env = _drone_batch(params_1k, 3, kahan_position=True)
    env.reset()
    a = torch.zeros((1000, 3, 4), dtype=torch.float32, device=DEV)
    for _ in range(10):
        env.rollout(a)
    torch.cuda.synchronize()
    got = env.state.cpu().numpy()
    ref = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(1, 9), g["prev_rates"][:, -1],
                          g["prev_thrust"][:, -1:]], axis=1)
    err = soa_vs_oracle(got[:, :1].copy(), ref, 1)
    assert_parity(err, REL_TOL, "config 1 @10k with Kahan rows")
    assert err["pos_rel"] < 1e-6, err
    model = lane_model.initial_state(params_1k, 3)
    comp = np.zeros((6, model.shape[1]), dtype=np.float32)
    lane_model.set_pos_comp(comp)
    try:
        lane_model.run(params_1k, model, np.zeros((3, 4), np.float32), steps=10000)
    finally:
        lane_model.set_pos_comp(None)
    assert np.array_equal(got[:, :3].view(np.uint32), model[:, :3].view(np.uint32))
    assert np.array_equal(env.pos_comp.cpu().numpy()[:, :3].view(np.uint32), comp[:, :3].view(np.uint32))
    env.reset(mask=np.array([1, 0, 0], dtype=np.uint8))
    torch.cuda.synchronize()
    pc = env.pos_comp.cpu().numpy()
    assert np.all(pc[:, 0] == 0) and np.any(pc[:, 1] != 0)


def test_config0_default_fps60_10k_steps_with_kahan_rows(params_60):
    """BASELINE configs[0] to the letter - params.yaml defaults (fps = 60), zero sticks, 10 000 steps - on the GPU with
    the Kahan rows, against the reference capture G1 @ fps 60 at every 100th step (1e-5; measured < 1e-6) and the
    host lane model bit for bit."""
    g = load_golden("g1_zero_10k_fps60")
    env = _drone_batch(params_60, 2, kahan_position=True)
    env.reset()
    a = torch.zeros((100, 2, 4), dtype=torch.float32, device=DEV)
    worst = 0.0
    for k, t in enumerate(np.asarray(g["snap_steps"]).reshape(-1)):
        env.rollout(a)
        ref = np.concatenate([g["state"][:, k], g["R"][:, k].reshape(1, 9), g["prev_rates"][:, k], g["prev_thrust"][:, k:k + 1]], axis=1)
        err = soa_vs_oracle(env.state.cpu().numpy()[:, :1].copy(), ref, 1)
        worst = max(worst, err["pos_rel"], err["pos_comp"], err["quat_abs"])
        assert int(t) == 100 * (k + 1)
    assert worst < REL_TOL, worst
    assert worst < 1e-6, worst
    assert not bool(env.done.any())
    model = lane_model.initial_state(params_60, 2)
    comp = np.zeros((6, model.shape[1]), dtype=np.float32)
    lane_model.set_pos_comp(comp)
    try:
        lane_model.run(params_60, model, np.zeros((2, 4), np.float32), steps=10000)
    finally:
        lane_model.set_pos_comp(None)
    assert np.array_equal(env.state.cpu().numpy()[:, :2].view(np.uint32), model[:, :2].view(np.uint32))


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


def test_checkpoint_does_not_depend_on_the_row_stride(params_1k):
    """fpv_recommended_ld changed between rounds (2^19 drones: n + 256 -> n + 320 floats; 10^6 drones: n -> n + 192) and
    differs between devices, so a checkpoint stores LOGICAL columns - and a checkpoint of rounds <= 5 (padded tensors with the
    writer's stride, flat fp16 words) still loads: the stride is read off the tensor.  Both continue bit for bit.  fp16 state: an
    ABI-5 file (the same encoding, written before checkpoints were labelled) loads with a warning; a labelled file of another
    encoding is refused; float16 sticks are cast the same way by step(), rollout() and step_async()."""
    import warnings
    from fpyv_amd.env import DroneBatch, FpvVecEnv
    n = 3000
    p = params_1k.replace(ceiling=10.3, noise_gain=2.0)
    for fp16 in (False, True):
        # (in-kernel stick noise needs fp32 state: the fp16 batch takes its sticks from a tensor)
        kw = dict(device=DEV, stick_noise=not fp16, noise_seed=4, auto_reset=True, with_accel=False, kahan_position=not fp16, fp16_state=fp16, rounding_seed=9)
        a, b, c = (DroneBatch(p, n, **kw) for _ in range(3))
        for e in (a, b, c):
            e.reset()
        gs = torch.Generator(device=DEV); gs.manual_seed(3)
        sticks_t = None if not fp16 else torch.rand((100, n, 4), device=DEV, generator=gs) * 2 - 1
        roll = (lambda e, t0, k: e.rollout(None, steps=k)) if not fp16 else (lambda e, t0, k: e.rollout(sticks_t[t0:t0 + k].contiguous()))
        roll(a, 0, 60)
        ck = a.state_dict()
        assert ck["layout"] == "columns" and ck["ld"] == a.ld and ck["state"].shape == (a.state.shape[0], n) and (fp16 or ck["noise_state"].shape == (4, n))
        if fp16:
            assert ck["state_h"].shape == (11, n) and ck["state_h"].dtype == torch.int16
            assert torch.equal(ck["state_h"], a.storage_words()[:, :n])
        # the same checkpoint as a library with ANOTHER row stride wrote it in rounds <= 5: padded tensors, flat fp16 words
        ld2 = a.ld + 448
        old = {k: v for k, v in ck.items() if k not in ("layout", "ld")}
        for k in ("state", "noise_state", "pos_comp"):
            if k in ck:
                t = torch.zeros((ck[k].shape[0], ld2), dtype=ck[k].dtype, device=DEV)
                t[:, :n] = ck[k]
                old[k] = t
        if fp16:
            w = torch.zeros(11 * ld2, dtype=torch.int16, device=DEV)
            w[:10 * ld2].view(5, ld2, 2)[:, :n] = ck["state_h"][:10].view(5, 2, n).permute(0, 2, 1)
            w[10 * ld2:10 * ld2 + n] = ck["state_h"][10]
            old["state_h"] = w.view(torch.float16)
        b.load_state_dict(ck)
        c.load_state_dict(old)
        for e in (a, b, c):
            roll(e, 60, 40)
        torch.cuda.synchronize()
        for k in ("state", "state_h", "noise_state", "pos_comp", "reward", "done_u8"):
            x = getattr(a, k, None)
            if x is not None:
                view = (lambda t: t.view(torch.int16)) if k == "state_h" else (lambda t: t)
                assert torch.equal(view(x), view(getattr(b, k))) and torch.equal(view(x), view(getattr(c, k))), (fp16, k)
        if fp16:
            abi5 = {k: v for k, v in old.items() if k not in ("state_h_encoding", "abi_version")}
            with pytest.warns(RuntimeWarning, match="ABI-5"):
                c.load_state_dict(abi5)
            with pytest.raises(ValueError, match="storage encoding"):
                c.load_state_dict(dict(old, state_h_encoding="abi3: eleven half rows"))
            with pytest.raises(ValueError, match="storage encoding"):
                c.load_state_dict({k: v for k, v in ck.items() if k != "state_h_encoding"})        # ABI >= 6 always labels
        with pytest.raises(ValueError, match="shape"):
            c.load_state_dict(dict(ck, state=ck["state"][:, :n - 1]))
    # one rule for sticks of another floating dtype: cast (warned about once), whichever call takes them
    e1, e2 = DroneBatch(p, 512, device=DEV), DroneBatch(p, 512, device=DEV)
    e1.reset(); e2.reset()
    g = torch.Generator(device=DEV); g.manual_seed(1)
    acts = (torch.rand((6, 512, 4), device=DEV, generator=g) * 2 - 1).half()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for t in range(6):
            e1.step(acts[t], return_imu=False)
        e2.rollout(acts)
        v = FpvVecEnv(p, num_envs=512, device=DEV, partitions=2)
        v.reset()
        for t in range(6):
            for part in range(v.partitions):
                lo, hi = v.partition_range(part)
                v.step_async(part, acts[t, lo:hi])
        for part in range(v.partitions):
            v.step_wait(part)
    torch.cuda.synchronize()
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.state, v.batch.state)
    v.close()


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


def test_feature_combinations_fuzz_bitwise(params_1k):
    """Every combination of the independent switches (auto-reset, ground flag | object list, Kahan
    rows, block width, ragged n) must select a kernel instantiation whose result equals the host
    lane model bit for bit."""
    rng = np.random.default_rng(2025)
    objs = ((2, 0.3, -0.2, 0.9, 0.35, 0.0), (1, 1.2, 0.4, 0.0, 0.5, 1.1), (0, 0, 0, 0, 0, 0))
    base = params_1k.replace(init_position=np.array([0.0, 0.0, 0.55]), ceiling=1.6)
    for case in range(24):
        auto, kahan = bool(case & 1), bool(case & 2)
        world = ("none", "flag", "list")[case % 3]
        n = int(rng.integers(1, 700))
        steps = int(rng.integers(5, 60))
        p = base.replace(ground=(world == "flag"))
        acts = rng.uniform(-1, 1, (steps, n, 4)).astype(np.float32)
        acts[..., 3] = rng.uniform(-1, -0.3, (steps, n))            # mostly below hover: ground/objects get hit
        pos = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(0.3, 1.2, (n, 1))], axis=1).astype(np.float32)
        env = _drone_batch(p, n, auto_reset=auto, kahan_position=kahan, with_done_bits=True)
        model = lane_model.initial_state(p, n, pos, [0.5, 0, 0], [0, 0, 0])
        env.state[:, :n] = torch.from_numpy(model[:, :n]).to(DEV)
        comp = np.zeros((6, model.shape[1]), dtype=np.float32)
        a = torch.from_numpy(acts).to(DEV)
        try:
            lane_model.set_pos_comp(comp if kahan else None)
            lane_model.set_objects(objs if world == "list" else ())
            for t in range(steps):
                env.step(a[t], object_list=objs if world == "list" else (), return_imu=False)
            _, acc, done, rew = lane_model.run(p, model, acts, auto_reset=auto)
        finally:
            lane_model.set_pos_comp(None)
            lane_model.set_objects(())
        torch.cuda.synchronize()
        tag = f"case {case}: auto={auto} kahan={kahan} world={world} n={n} steps={steps}"
        assert np.array_equal(env.state.cpu().numpy()[:, :n].view(np.uint32), model[:, :n].view(np.uint32)), tag
        assert np.array_equal(env.done_u8.cpu().numpy(), done), tag
        assert np.array_equal(env.reward.cpu().numpy().view(np.uint32), rew.view(np.uint32)), tag
        if kahan:
            assert np.array_equal(env.pos_comp.cpu().numpy()[:, :n].view(np.uint32), comp[:, :n].view(np.uint32)), tag


def test_big_angle_path_on_gpu(params_1k):
    """max_rates so large that one step can turn more than 90 degrees: fpv_create selects angle mode 2 (range
    reduction without a library call, fpv_sincos_reduced); same physical trajectory as the small-angle kernel and the
    oracle, and - new in round 3 - the same bits as the host build."""
    g = load_golden("g3_ema_noise")
    n = g["actions"].shape[1]
    p_big = params_1k.replace(max_rates=2.0e5)
    acts = (g["actions"] * np.float32(1e-3)).astype(np.float32)
    acts[..., 3] = g["actions"][..., 3]
    env = _drone_batch(p_big, n)
    env.reset()
    env.rollout(torch.from_numpy(acts).to(DEV))
    torch.cuda.synchronize()
    ref = oracle.drone_initial_state(n, p_big.init_position, p_big.init_velocity, [0, 0, 0])
    oracle.drone_run(p_big, ref, acts.astype(np.float64))
    assert_parity(soa_vs_oracle(env.state.cpu().numpy(), ref, n), 2e-5, "big-angle kernel")
    # genuinely large per-step rotations (tumbling at 40 000 deg/s with dt = 1 ms = 80 deg per application)
    acts2 = np.zeros((200, n, 4), dtype=np.float32)
    acts2[..., 0] = 0.2; acts2[..., 1] = -0.15; acts2[..., 3] = -0.5
    env.reset()
    env.rollout(torch.from_numpy(acts2).to(DEV))
    torch.cuda.synchronize()
    ref = oracle.drone_initial_state(n, p_big.init_position, p_big.init_velocity, [0, 0, 0])
    oracle.drone_run(p_big, ref, acts2.astype(np.float64))
    err = soa_vs_oracle(env.state.cpu().numpy(), ref, n)
    assert err["quat_abs"] < 5e-5 and err["pos_rel"] < 5e-5, err
    model = lane_model.initial_state(p_big, n)
    lane_model.run(p_big, model, acts2)
    assert np.array_equal(env.state.cpu().numpy()[:, :n].view(np.uint32), model[:, :n].view(np.uint32)), "big-angle kernel != lane model (bitwise)"
    single = _drone_batch(p_big, n)
    single.reset()
    a2 = torch.from_numpy(acts2).to(DEV)
    for t in range(acts2.shape[0]):
        single.step(a2[t], return_imu=False)
    torch.cuda.synchronize()
    assert torch.equal(single.state, env.state), "single-step and k-step kernels must agree in angle mode 2 as well"


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


# ---- fpv_step_n: k steps in ONE launch, bit-identical to k single steps ---------------------------------
def _clone_batch_state(dst, src):
    for name in ("state", "state_h", "noise_state", "pos_comp", "ep_return", "ep_length", "last_return", "last_length"):
        a, b = getattr(dst, name, None), getattr(src, name, None)
        if a is not None:
            a.copy_(b)


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


def test_obs_aos_rows_vs_reference_return_triple(params_1k):
    """The AoS observation row against the reference capture directly: p, v from `state`, q against R,
    prev_rates, and the accelerometer triple R_new @ acc (components.py:247-248) of golden G3."""
    g = load_golden("g3_ema_noise")
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(params_1k, n, with_obs_aos=True)
    env.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    a = torch.from_numpy(acts).to(DEV)
    for t in range(T):
        env.step(a[t], return_imu=False)
    torch.cuda.synchronize()
    obs = env.obs_aos.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(obs[:, 0:3], g["state"][:, -1, 0:3], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(obs[:, 3:6], g["state"][:, -1, 3:6], rtol=2e-5, atol=2e-5)
    Rq = oracle.quat_to_matrix(obs[:, 6:10])
    assert np.abs(Rq - g["R"][:, -1]).max() < 2e-5
    np.testing.assert_allclose(obs[:, 10:13], g["prev_rates"][:, -1], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(obs[:, 13:16], g["accel"][:, -1], rtol=1e-4, atol=1e-4)


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


def _two_host_threads_two_handles(params, devices, one_world):
    """Two host threads in ONE process, a handle and a communicator rank per thread, neither thread ever calling
    hipSetDevice itself.  one_world: both threads join ONE communicator of world size 2 (needs two GPUs); otherwise each
    thread has its own one-rank communicator (what a one-GPU box can run: the threading of the C ABI - thread-local error
    strings, RCCL opened under call_once - and the device guard are the same code)."""
    import ctypes as C
    import threading
    L = _lib.lib()
    idents = []
    for _ in range(1 if one_world else 2):
        ident = (C.c_uint8 * _lib.FPV_COMM_ID_BYTES)()
        _lib.check(L.fpv_comm_unique_id(ident))
        idents.append(ident)
    torch.cuda.set_device(0)
    n, k = 5000, 16
    words = (n + 63) // 64
    out, errors = {}, []
    p = params.replace(ceiling=10.0005)
    world = 2 if one_world else 1

    def rank_main(r):
        try:
            dev = torch.device("cuda", devices[r])
            seen = [torch.cuda.current_device()]                      # a fresh thread: device 0 is current, also for rank 1
            from fpyv_amd.env import DroneBatch
            env = DroneBatch(p, n, device=dev, with_done_bits=True, auto_reset=True, with_accel=False, drone_id_offset=r * n)
            env.reset()
            acts = torch.from_numpy(sticks.ema_noise(k, range(r * n, (r + 1) * n), seed=4)).to(dev)
            acts[..., 3] = 1.0                                          # full throttle: through the ceiling within a few steps
            if r == 1:
                acts[:, ::3, 3] = -0.9                                  # every third drone of rank 1 sinks instead: the ranks' masks differ
            comm = C.c_void_p()
            _lib.check(L.fpv_comm_create(idents[0 if one_world else r], world, r if one_world else 0, devices[r], C.byref(comm)))   # collective
            seen.append(torch.cuda.current_device())
            ws, rk, ver = C.c_int(-1), C.c_int(-1), C.c_int(-1)
            _lib.check(L.fpv_comm_info(comm, C.byref(ws), C.byref(rk), C.byref(ver)))
            bucket = torch.zeros((k, words), dtype=torch.int64, device=dev)
            dones = torch.zeros((k, n), dtype=torch.uint8, device=dev)
            env.set_done_bits_target(bucket, stride_words=words)
            env.rollout(acts, dones=dones)                              # ONE launch on this rank's GPU writes all k mask rows
            seen.append(torch.cuda.current_device())
            gathered = torch.full((world, k, words), -1, dtype=torch.int64, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.fpv_allgather_done(comm, bucket.data_ptr(), gathered.data_ptr(), k * words, stream))
            seen.append(torch.cuda.current_device())
            torch.cuda.synchronize(dev)
            assert L.fpv_allgather_done(comm, None, gathered.data_ptr(), words, stream) == -1
            assert b"null argument" in L.fpv_last_error()            # this thread's own message (thread-local)
            out[r] = dict(bucket=bucket.cpu(), gathered=gathered.cpu(), dones=dones.cpu(), seen=seen, info=(ws.value, rk.value, ver.value),
                          state_device=env.state.device.index)
            L.fpv_comm_destroy(comm)
            seen.append(torch.cuda.current_device())
        except Exception as e:      # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in RCCL"
    assert not errors, errors
    assert torch.cuda.current_device() == 0
    whole = torch.stack([out[0]["bucket"], out[1]["bucket"]])
    for r in range(2):
        o = out[r]
        assert o["info"][:2] == ((2, r) if one_world else (1, 0)) and o["info"][2] >= 20000
        assert o["seen"] == [0, 0, 0, 0, 0], f"rank {r}: an fpv_* call left the caller's current device changed: {o['seen']}"
        assert o["state_device"] == devices[r]
        assert torch.equal(o["gathered"], whole if one_world else whole[r:r + 1]), f"rank {r}: gathered masks != concatenation of the ranks' buckets"
        bits = ((o["bucket"].numpy().view(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(k, -1)[:, :n]
        assert np.array_equal(bits.astype(np.uint8), o["dones"].numpy()) and bits.any()
    assert not torch.equal(out[0]["bucket"], out[1]["bucket"])


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


def test_guidance_override_vs_reference_capture(params_1k):
    """Drone.step(action, wind, object_list, rotation_matrix=R, thrust_force=f) - the guidance call of
    simulator.py:110 (components.py:230-232) - through the public API against the reference capture G13: override
    switched on and off mid-flight (NaN thrust_force = that drone is not overridden), on every step, and on
    every step above a Ground object.  Bitwise against the host build of the same arithmetic, 1e-5 against the
    reference, `done` equal on every step."""
    from fpyv_amd.objects import Ground
    g = load_golden("g13_guidance_override")
    acts = g["actions"]
    T = acts.shape[0]
    for cases, world, rows in (([0, 1], [], ()), ([2], [Ground()], [(0, 0, 0, 0, 0, 0)])):
        m = len(cases)
        env = _drone_batch(params_1k, m)
        model = lane_model.initial_state(params_1k, m, g["init_position"][cases], g["init_velocity"][cases], g["init_ypr"][cases])
        env.state[:, :m] = torch.from_numpy(model[:, :m]).to(DEV)
        a = torch.from_numpy(np.ascontiguousarray(acts[:, cases])).to(DEV)
        R = torch.from_numpy(np.ascontiguousarray(g["rotation_override"][:, cases]).astype(np.float32)).to(DEV)
        f = torch.from_numpy(np.ascontiguousarray(g["thrust_force"][:, cases]).astype(np.float32)).to(DEV)
        lane_model.set_objects(rows)
        try:
            for t in range(T):
                ret = env.step(a[t], wind_velocity_vector=np.zeros(3), object_list=world, rotation_matrix=R[t], thrust_force=f[t])
                assert np.array_equal(env.done_u8.cpu().numpy(), g["done"][cases, t])
                lane_model.set_override(g["rotation_override"][t, cases], g["thrust_force"][t, cases])
                lane_model.run(params_1k, model, acts[t:t + 1, cases])
        finally:
            lane_model.set_override(None)
            lane_model.set_objects(())
        got = env.state.cpu().numpy()
        assert np.array_equal(got[:, :m].view(np.uint32), model[:, :m].view(np.uint32)), "kernel != lane model (bitwise)"
        ref = np.concatenate([g["state"][cases, -1], g["R"][cases, -1].reshape(m, 9), g["prev_rates"][cases, -1],
                              g["prev_thrust"][cases, -1][:, None]], axis=1)
        assert_parity(soa_vs_oracle(got, ref, m), REL_TOL, f"G13 cases {cases}")
        # the return triple of the last (overridden) step: R_new.T and R_new @ acc
        np.testing.assert_allclose(ret[0].cpu().numpy(), g["ret_RT"][cases], atol=2e-6)
        np.testing.assert_allclose(ret[2].cpu().numpy(), g["accel"][cases, -1], rtol=1e-4, atol=1e-4)
    # a [3,3] matrix and a scalar force broadcast over the batch; the override is a per-step input
    env = _drone_batch(params_1k, 5)
    env.reset()
    env.step(np.zeros(4, np.float32), rotation_matrix=np.eye(3), thrust_force=7.0)
    env.step(np.zeros(4, np.float32), thrust_force=7.0)                    # ignored without rotation_matrix (components.py:230)
    with pytest.raises(TypeError):
        env.step(np.zeros(4, np.float32), rotation_matrix=np.eye(3))
    with pytest.raises(ValueError):
        env.step(np.zeros(4, np.float32), rotation_matrix=np.zeros((4, 3, 3)), thrust_force=1.0)
    assert env._buf.rotation_override is None and env._buf.thrust_override is None, "the override must not outlive its step"


@pytest.mark.parametrize("seed", range(4))
def test_random_drone_types_bitwise_and_1e5(params_1k, seed):
    """Drone types far from params.yaml (tests/parity.py::random_drone_params: mass, X or rectangular frame, thrust
    curve, drag, rates, low-pass constants, gravity, dt): the single-step kernel and the k-step kernel (which
    picks the two-height ground flag only for the X frame) equal the host build bit for bit and hold 1e-5 against
    the float64 oracle after 1000 steps."""
    from parity import assert_parity_random_type, random_drone_params
    rng = np.random.default_rng(1000 + seed)
    p = random_drone_params(params_1k, rng)
    n, steps = 333, 1000
    acts = sticks.ema_noise(steps, range(n), seed=seed)
    acts[..., 3] += np.float32(rng.uniform(-0.7, -0.3))
    model = lane_model.initial_state(p, n)
    a = torch.from_numpy(acts).to(DEV)
    single, fused = _drone_batch(p, n), _drone_batch(p, n)
    for e in (single, fused):
        e.state[:, :n] = torch.from_numpy(model[:, :n]).to(DEV)
    single.rollout(a, fused=False)
    fused.rollout(a)
    lane_model.run(p, model, acts)
    got = single.state.cpu().numpy()
    assert np.array_equal(got[:, :n].view(np.uint32), model[:, :n].view(np.uint32)), "kernel != lane model (bitwise)"
    assert torch.equal(single.state, fused.state) and torch.equal(single.done_u8, fused.done_u8)
    ref = oracle.drone_initial_state(n, p.init_position, p.init_velocity, p.init_orientation_deg)
    oracle.drone_run(p, ref, acts.astype(np.float64))
    assert_parity_random_type(got, ref, n, p, REL_TOL, f"random drone type {seed}")


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


def test_stream_probe_and_busy_kernel_surface():
    """fpyv_amd.streams: the choice of the split-phase API's partition streams.  fpv_diag_busy is a kernel of known duration on one
    CU: it refuses durations outside (0, 1000] us, launches on the caller's stream and completes; the probe hands out two distinct
    streams that are not the caller's.  (How long the busy kernels take and whether the chosen streams overlap are wall-clock
    questions: tests/test_gpu_timing.py, marker gpu_timing - not part of the parity gate.)"""
    from fpyv_amd.streams import overlapping_streams
    L = _lib.lib()
    assert L.fpv_diag_busy(0.0, None) == -1 and L.fpv_diag_busy(2000.0, None) == -1 and b"microseconds" in L.fpv_last_error()
    s = torch.cuda.Stream(device=DEV)
    for _ in range(20):
        _lib.check(L.fpv_diag_busy(200.0, s.cuda_stream))
    s.synchronize()
    cur = torch.cuda.current_stream(DEV)
    picked, rep = overlapping_streams(DEV, 2, avoid=[cur])
    assert len(picked) == 2 and picked[0] != picked[1] and cur not in picked
    assert isinstance(rep["verified"], bool) and rep["draws"] >= 1 and len(rep["ratios"]) >= 1


@pytest.mark.parametrize("extra", [[], ["--partitions", "2"], ["--api", "rollout"]], ids=["step", "partitions2", "rollout"])
def test_bench_line_schema_small(extra):
    """bench.py end to end at a small size: ONE JSON line with the contract's keys, `roofline` (achieved / peak / frac / traffic /
    sustained leg for the step API) and - N = 1, step API - `cpu_baseline`; the split-phase line says how its streams were chosen;
    the k-step line without a counted instruction mix for this size keeps bound = "hbm" and says why."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "40", "--warmup", "8", "--drones-per-gpu", str(1 << 16),
                        "--sustained-steps", "64", "--no-beyond-mall", "--preheat-s", "0.05", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8 and d["unit"] == "env-steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - (1 << 16) * 40 / (d["ms_per_step"] * 40e-3)) < 1e-3 * d["value"]        # consistent with its own clock; how fast is not the gate's business
    ro = d["roofline"]
    assert ro["peak"] == 8000.0 and 0 < ro["frac"] and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and ro["traffic"] is None   # traffic is quoted for the headline size only
    # one line, one answer: frac follows `value` by the stated formula; the HIP-event figure stands beside it under its own name
    hv = ro["hbm_view"] if extra[:1] == ["--api"] else ro
    assert abs(hv["frac"] - d["value"] / d["n_gpus"] * ro["algorithmic_bytes_per_env_step"] / 1e9 / 8000.0) < 1e-9 * max(1.0, hv["frac"])
    assert hv["frac_events"] > 0 and "value" in ro["frac_formula"]
    if extra[:1] == ["--api"]:
        assert ro["bound"] == "hbm" and ro["valu"] is None and ("configuration" in ro["valu_unavailable"] or "stale" in ro["valu_unavailable"])   # (stale: sources edited since the counter pass)
        assert d["config"]["steps_per_launch"] > 1
    else:
        assert ro["bound"] == "hbm" and ro["sustained"]["launches"] == 64 and ro["sustained"]["avg_launch_us"] > 0
        assert d["config"]["partitions"] == (2 if extra else 1)
        if extra:
            assert d["config"]["partition_streams"]["verified"] is True


def test_bench_line_auxiliary_legs_at_the_headline_size():
    """The legs only the full-size line has: `beyond_mall` at 2^23 drones with its three repeats, host enqueue
    time and buffer addresses, and `launch_time_fit` over 3 * 2^18 / 2^20 / 2^21 drones (beyond the L2s, inside the Infinity Cache) with per-leg repeats, host enqueue time
    and a verdict on its own validity: the line must be consistent with that verdict (what the numbers ARE is the bench line's
    business, not the parity gate's)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--sustained-steps", "200", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    ro = json.loads(lines[0])["roofline"]
    assert ro["host_enqueue_us"] > 0 and isinstance(ro["host_bound"], bool) and ro["sustained"]["host_enqueue_us"] > 0
    assert ro["host_bound"] == (ro["host_enqueue_us"] > 0.9 * ro["avg_launch_us"])
    b = ro["beyond_mall"]
    assert b["drones"] == 1 << 23 and len(b["repeats_us"]) == 3 and b["launches_per_repeat"] == 100 and b["host_enqueue_us"] > 0
    assert all(t > 0 for t in b["repeats_us"]) and b["frac"] > 0 and b["frac_of_copy_ceiling"] > 0 and set(b["addresses"])   # how large: the line itself says (bench.py), no gate
    assert set(b["addresses"]) == {"state", "ld", "action", "reward", "done"}
    lf = ro["launch_time_fit"]
    assert lf["drones"] == [3 << 18, 1 << 20, 1 << 21] and len(lf["legs"]) == 3
    for leg in lf["legs"]:
        assert len(leg["repeats_us"]) == 3 and leg["launches"] == 400 and leg["host_enqueue_us"] > 0 and leg["avg_launch_us"] > 0
        assert leg["host_bound"] == (leg["host_enqueue_us"] > 0.9 * leg["avg_launch_us"])       # a slow host is SAID, not asserted away
    # the fit judges itself: on a warm, quiet GPU it is valid (floor of a few microseconds); a box on which a leg is off the line
    # must say so instead of printing a floor - either way the line is consistent with its own verdict
    if lf["valid"]:
        assert lf["invalid_reason"] is None and lf["floor_us"] > 0 and lf["max_residual_us"] <= 0.5 and 0.0 < lf["streaming_frac_of_peak"] <= 1.0    # (fit_launch_time's own validity rules)
        assert abs(lf["floor_share_of_headline_launch"] - lf["floor_us"] / ro["sustained"]["avg_launch_us"]) < 1e-9
    else:
        assert lf["invalid_reason"] and lf["floor_share_of_headline_launch"] is None


@pytest.mark.parametrize("kind", ["plain", "noise", "objects", "kahan", "guidance", "fp16", "aos", "racer", "racer_written"])
def test_rotation_of_the_traversal_is_bit_identical(params_1k, kind):
    """fpv_set_rotation (ABI 7): the fp32 drone step kernels start `drones` before the previous launch's start and wrap - so that a
    population beyond the 256 MiB Infinity Cache begins each launch on the rows it wrote last.  The ORDER of the workgroups must
    not matter: every buffer bit for bit the plain order's, for every instantiation, ragged n (a partial last block), steps through
    step(), rollout(fused=False) and the hipGraph replay, a rotation smaller and larger than the population."""
    from fpyv_amd.env import DroneBatch, RacerBatch
    from fpyv_amd.objects import Cylinder, Ground
    n, T = 70001, 24
    p = params_1k.replace(ceiling=10.2, init_position=np.array([0.0, 0.0, 0.03]), init_velocity=np.array([1.0, 0.2, -3.0]))   # through z = 0 within 10 ms
    kw = dict(device=DEV, auto_reset=True, with_accel=True, with_done_bits=True, track_episodes=True)
    objs = ()
    racer = kind.startswith("racer")
    if racer:
        pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0]])
        p = params_1k.replace(mode=1, racer_pid=pid, racer_omega_dt=(kind == "racer"), ceiling=2e-4)
        kw.pop("with_accel")
    if kind == "fp16":
        kw.update(fp16_state=True, rounding_seed=3, with_accel=False)
    if kind == "aos":
        kw.update(with_obs_aos=True)
    if kind == "noise":
        kw.update(stick_noise=True, noise_seed=5, with_action_out=True)
    if kind == "kahan":
        kw.update(kahan_position=True)
    if kind == "objects":
        objs = [Ground(), Cylinder(position=[1.0, 0.2, 0.0], radius=0.5, height=1.0)]
    g = torch.Generator(device=DEV); g.manual_seed(4)
    acts = torch.rand((T, n, 4), device=DEV, generator=g) * 2 - 1
    R = torch.eye(3, device=DEV).expand(n, 3, 3).contiguous()
    thrust = torch.full((n,), 6.0, device=DEV)

    def run(rotation):
        e = (RacerBatch if racer else DroneBatch)(p, n, **kw)
        assert e.rotation == 0, "a population that fits the cache keeps the plain order by default"
        e.set_rotation(rotation)
        assert e.rotation == (rotation // 128 * 128) % ((n + 1023) // 1024 * 1024) or rotation == 0      # blocks in whole rounds of the eight XCDs
        e.reset()
        ol = {} if racer else dict(object_list=objs)
        for t in range(8):
            if kind == "guidance":
                e.step(acts[t], rotation_matrix=R, thrust_force=thrust, return_imu=False)
            elif racer:
                e.step(acts[t] * torch.tensor([3.0, 3.0, 3.0, 4.0], device=DEV))
            else:
                e.step(acts[t], object_list=objs, return_imu=False)
        e.rollout(acts[8:16], fused=False, **ol)
        e.rollout(acts[16:24], graph=True, **ol)
        e.rollout(acts[16:24], graph=True, **ol)                                   # a replay of the cached graph
        torch.cuda.synchronize()
        return e

    base = run(0)
    for rotation in (384, 128 * 300, 128 * 9000):
        other = run(rotation)
        for name in ("state", "state_h", "reward", "done_u8", "done_bits", "accel", "ep_return", "ep_length", "last_return", "last_length", "noise_state", "action_out",
                     "pos_comp", "obs_aos"):
            x, y = getattr(base, name, None), getattr(other, name, None)
            if x is not None:
                x, y = (x.view(torch.int16), y.view(torch.int16)) if name == "state_h" else (x, y)
                assert torch.equal(x, y), (kind, rotation, name)
    assert int(base.last_length.max()) > 0, "the run must end episodes (auto-reset inside the rotated order too)"


def test_rotation_is_automatic_beyond_the_l2s_and_beyond_the_infinity_cache(params_1k):
    """The automatic rule (fpv_abi.h): plain order while what one launch writes fits 61/64 of the eight L2s; beyond them the start
    moves back by the L2s' share of drones per launch (2^19 for the plain kernel's 61 B), beyond the 256 MiB Infinity Cache by its
    share (2^22; fewer with the four noise rows, accel rows, Kahan rows).  At 2^23 and at 2^20 drones the automatically rotated
    chain leaves the same bits as the plain order on the same buffers.  (What the rotation is worth in time:
    tests/test_gpu_timing.py and bench.py's `beyond_mall.plain_order_avg_launch_us` - not a parity question.)"""
    from fpyv_amd import sticks
    from fpyv_amd.env import DroneBatch
    share = lambda cache, written: cache // 64 * 61 // written // 128 // 8 * 8 * 128      # noqa: E731  (61/64 of the cache, whole rounds of the eight XCDs)
    L2, MALL = 32 << 20, 256 << 20
    assert DroneBatch(params_1k, 1 << 19, device=DEV, with_accel=False).rotation == 0                      # a launch writes 32 MB: the L2s hold it
    assert DroneBatch(params_1k, 1 << 20, device=DEV, with_accel=False).rotation == share(L2, 61) == 1 << 19
    assert DroneBatch(params_1k, 1 << 22, device=DEV, with_accel=False).rotation == 1 << 19
    assert DroneBatch(params_1k, (1 << 22) + 128, device=DEV, with_accel=False).rotation == share(MALL, 61) == 1 << 22
    noisy = DroneBatch(params_1k, 5 << 20, device=DEV, with_accel=False, stick_noise=True)
    assert noisy.rotation == share(MALL, 61 + 16)
    assert DroneBatch(params_1k, 5 << 20, device=DEV, with_accel=False, fp16_state=True).rotation == share(L2, 39)      # 39 B written per drone: 5 M drones fit the Infinity Cache
    assert DroneBatch(params_1k, 8 << 20, device=DEV, with_accel=False, fp16_state=True).rotation == share(MALL, 39)
    acc = DroneBatch(params_1k, 1 << 20, device=DEV, with_accel=True, kahan_position=True)               # what a launch writes decides: + accel rows + Kahan rows
    assert acc.rotation == 1 << 19                                                                       # (the estimate before the first launch knows reward and done only)
    acc.reset(); acc.step(torch.zeros((1 << 20, 4), device=DEV), return_imu=False)
    assert acc.rotation == share(L2, 61 + 24)                                                            # (the accel rows leave with a streaming hint and are not counted; the Kahan rows are re-read)
    aos = DroneBatch(params_1k, 1 << 20, device=DEV, with_accel=False, with_obs_aos=True)                # the AoS head likewise: written once, streamed
    aos.reset(); aos.step(torch.zeros((1 << 20, 4), device=DEV), return_imu=False)
    assert aos.rotation == 1 << 19
    del aos
    del acc
    del noisy
    torch.cuda.empty_cache()
    for n, ring, want in ((1 << 23, 4, 1 << 22), (1 << 20, 16, 1 << 19)):
        e = DroneBatch(params_1k.replace(ceiling=100.0), n, device=DEV, auto_reset=True, with_accel=False)
        assert e.rotation == want
        acts = sticks.ema_noise_device(ring, n, DEV, seed=9)

        def final():
            e.reset()
            for _ in range(3):
                e.rollout(acts, fused=False)
            torch.cuda.synchronize()
            return e.state.clone()

        s_rot = final()
        e.set_rotation(0)
        assert e.rotation == 0
        assert torch.equal(s_rot, final())
        del e, acts, s_rot
        torch.cuda.empty_cache()


def test_ragged_population_runs_in_whole_rounds_of_the_xcds(params_1k):
    """1 000 000 drones are 7812.5 blocks of 128: the traversal runs over whole rounds of the eight XCDs (7816 blocks, three of
    them empty) so that a block keeps its XCD across the wrap (profiles/r05_exp_row_stride_l2_sets.log sections 3-4).  Same results
    as the plain order, and equal to the sum of two batches that split the population at a block boundary."""
    from fpyv_amd import sticks
    from fpyv_amd.env import DroneBatch
    n = 1_000_000
    p = params_1k.replace(ceiling=100.0)
    e = DroneBatch(p, n, device=DEV, auto_reset=True, with_accel=False)
    assert e.rotation == 1 << 19 and e.ld == _lib.lib().fpv_recommended_ld(n) and e.ld % 512 == 256
    acts = sticks.ema_noise_device(16, n, DEV, seed=4)

    def final():
        e.reset()
        e.rollout(acts, fused=False)
        torch.cuda.synchronize()
        return e.state[:, :n].clone()

    s_rot = final()
    e.set_rotation(0)
    assert torch.equal(s_rot, final())
    cut = 499_968                                               # a block boundary: the two halves see the same sticks, drone for drone
    parts = [DroneBatch(p, m, device=DEV, auto_reset=True, with_accel=False) for m in (cut, n - cut)]
    for q, lo in zip(parts, (0, cut)):
        q.reset()
        q.rollout(acts[:, lo:lo + q.n].contiguous(), fused=False)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([q.state[:, :q.n] for q in parts], dim=1), s_rot)


def test_results_do_not_depend_on_the_row_stride(params_1k):
    """2^19 drones with the former pad of 256 floats and with the stride of fpv_recommended_ld (2 MiB + 1.25 KiB, chosen by the L2
    set model: profiles/r05_exp_row_stride_l2_sets.log): the same numbers in the same rows - results do not depend on ld."""
    import ctypes as C
    from fpyv_amd import sticks
    L = _lib.lib()
    n = 1 << 19
    rec = int(L.fpv_recommended_ld(n))
    assert rec == n + 320
    cp = _lib.pack_params(params_1k.replace(ceiling=100.0), auto_reset=True)
    h = C.c_void_p()
    assert L.fpv_create(C.byref(cp), n, 0, C.byref(h)) == 0
    acts = sticks.ema_noise_device(32, n, DEV, seed=2)
    rew, done = torch.zeros(n, device=DEV), torch.zeros(n, dtype=torch.uint8, device=DEV)
    big = torch.zeros(14 * (n + 512), device=DEV)
    finals = {}
    for ld in (n + 256, rec):
        st = big[:14 * ld].view(14, ld)
        b = _lib.FpvBuffers()
        b.state, b.ld, b.reward, b.done, b.action = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr(), acts.data_ptr()
        big.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
        assert L.fpv_set_step_counter(h, 0) == 0
        for _ in range(3):
            assert L.fpv_rollout(h, C.byref(b), 32, n * 4, 0, None) == 0
        torch.cuda.synchronize()
        finals[ld] = st[:, :n].clone()
    L.fpv_destroy(h)
    assert torch.equal(finals[n + 256], finals[rec])


def test_handle_lifecycle_does_not_leak(params_1k):
    """Create - use - destroy, a few hundred times: plain and noise handles, single steps, the k-step kernel, a cached hipGraph and its
    replay, a partitioned env with its streams, a PID handle.  Device memory outside torch's allocator (the handles' own tables,
    graphs, events) and the host's resident set must not grow with the count."""
    import gc
    import resource
    from fpyv_amd import sticks
    from fpyv_amd.env import DroneBatch, FpvVecEnv
    n = 4096
    acts = sticks.ema_noise_device(8, n, DEV, seed=1)

    def cycle(k):
        for i in range(k):
            e = DroneBatch(params_1k, n, device=DEV, auto_reset=True, stick_noise=(i % 2 == 1), noise_seed=i, with_done_bits=True, track_episodes=True)
            e.reset()
            a = None if i % 2 else acts
            e.step(None if i % 2 else acts[0], return_imu=False)
            e.rollout(a, steps=8)
            e.rollout(a, steps=8, graph=True) if i % 2 else e.rollout(acts, graph=True)
            e.rollout(a, steps=8, graph=True) if i % 2 else e.rollout(acts, graph=True)
            e.close()
            if i % 10 == 0:
                v = FpvVecEnv(params_1k, num_envs=n, device=DEV, partitions=2, auto_reset=True)
                v.reset()
                v.step(acts[1])
                v.close()
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()

    # the first few hundred lifecycles grow the runtime's own pools once (torch hands out its 32 pooled streams one after the other, and
    # HIP sets a stream up at its first use: 28 MiB in all); a leak would keep growing - so the SECOND window is the one that counts
    cycle(40)
    cycle(160)
    free1, rss1 = torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    cycle(160)
    free2, rss2 = torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free1 - free2 < (4 << 20), f"device memory shrank by {(free1 - free2) >> 20} MiB over 160 further handle lifecycles"
    assert rss2 - rss1 < (32 << 10), f"host resident set grew by {(rss2 - rss1) >> 10} MiB over 160 further handle lifecycles"       # ru_maxrss is in KiB
