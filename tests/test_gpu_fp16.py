"""GPU parity, BASELINE configs[3]: fp16 state storage / fp32 integrator (89 B per env-step) - bitwise against the host lane model,
the restated tolerance against the oracle, the storage format, shard invariance, the widening kernel."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from gpu_helpers import DEV, _drone_batch
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]


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
