"""GPU parity proper: the HIP path, called through the C ABI (fpyv_amd.env -> ctypes -> libfpv_hip.so), against the float64
oracle on the golden stick profiles, against the reference's own captured outputs, and - at BASELINE.json's full sizes - through
size-independent properties.  Tolerance (BASELINE.json north_star): 1e-5 relative on position / quaternion after 1000 steps.
(fp16 state: test_gpu_fp16.py; k-step kernels: test_gpu_kstep.py; the boundary: test_gpu_boundary.py; traversal order, row stride,
bench lines: test_gpu_traversal.py; wall-clock ratios, outside the gate: test_gpu_timing.py.)"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from gpu_helpers import DEV, _drone_batch, _run_golden, _racer_replay
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]


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
