"""Property tests (hypothesis) of the kernel's per-lane arithmetic on the host lane model: things
that must hold for ANY parameters / sticks, not just the golden profiles.  (The GPU suite repeats
the size-independent ones - unit quaternions, batch/lane/shard invariance - at 2^20 and 2^23 drones.)"""
import numpy as np
from hypothesis import given, settings, strategies as st

from fpyv_amd import load_params
from oracle import lane_model, oracle
from parity import soa_vs_oracle

P = load_params(fps=1000)
sticks_st = st.lists(st.floats(-1.5, 1.5, allow_nan=False, width=32), min_size=4, max_size=4)


@settings(max_examples=40, deadline=None)
@given(a=sticks_st, ypr=st.lists(st.floats(-179, 179), min_size=3, max_size=3),
       v0=st.lists(st.floats(-15, 15), min_size=3, max_size=3), steps=st.integers(1, 400),
       fps=st.sampled_from([60, 250, 1000]))
def test_unit_quaternion_finite_state_and_oracle_agreement(a, ypr, v0, steps, fps):
    p = P.replace(dt=1.0 / fps)
    act = np.array([a], dtype=np.float32)
    s = lane_model.initial_state(p, 1, [0, 0, 50.0], v0, ypr)
    lane_model.run(p, s, act, steps=steps)
    assert np.all(np.isfinite(s[:, 0]))
    assert abs(np.linalg.norm(s[6:10, 0].astype(np.float64)) - 1) < 5e-7
    assert np.all(np.abs(s[10:13, 0]) <= p.max_rates * (1 + 1e-6)), "low-passed rates stay inside +-max_rates"
    ref = oracle.drone_initial_state(1, [0, 0, 50.0], v0, ypr)
    oracle.drone_run(p, ref, act.astype(np.float64), steps=steps)
    err = soa_vs_oracle(s, ref, 1)
    # fp32 vs float64 over <= 400 steps from arbitrary attitudes and speeds.  Per component with a 1 m floor (`pos_comp`): the
    # drone starts at (0, 0, 50) and a powered fall takes it THROUGH the origin, where the error relative to |p| (`pos_rel`)
    # is a division by nearly nothing (seed 209 of the explored ones: 5.6e-5 at |p| = 0.04 m with pos_comp = 2.0e-6)
    # (the worst of 330 explored seeds: 2.03e-5 after a 5.7 s powered fall of 160 m at 60 fps - BASELINE's 1e-5 is for its configs)
    assert err["pos_comp"] < 3e-5 and err["quat_abs"] < 2e-5, err


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(1, 70), k=st.integers(0, 69))
def test_a_drone_does_not_depend_on_its_batch_or_lane(seed, n, k):
    k = k % n
    rng = np.random.default_rng(seed)
    acts = rng.uniform(-1, 1, (30, n, 4)).astype(np.float32)
    pos = rng.uniform(-5, 5, (n, 3)); pos[:, 2] += 20
    ypr = rng.uniform(-90, 90, (n, 3))
    s = lane_model.initial_state(P, n, pos, [1.0, 0, 0], ypr)
    lane_model.run(P, s, acts)
    alone = lane_model.initial_state(P, 1, pos[k], [1.0, 0, 0], ypr[k])
    lane_model.run(P, alone, np.ascontiguousarray(acts[:, k:k + 1]))
    assert np.array_equal(s[:, k].view(np.uint32), alone[:, 0].view(np.uint32))


@settings(max_examples=30, deadline=None)
@given(z=st.floats(-0.5, 0.5), roll=st.floats(-80, 80), pitch=st.floats(-80, 80))
def test_done_flag_is_exactly_any_motor_below_ground(z, roll, pitch):
    s = lane_model.initial_state(P, 1, [0, 0, z], [0, 0, 0], [roll, pitch, 0])
    R = oracle.quat_to_matrix(s[6:10, 0].astype(np.float64))[0]
    motors_z = np.float32(z) + (P.motor_xy @ R[2, :2])
    _, _, done, _ = lane_model.run(P, s, np.zeros((1, 4), np.float32), steps=1)
    if np.min(np.abs(motors_z)) > 1e-5:            # away from the fp32 decision boundary
        assert bool(done[0]) == bool((motors_z < 0).any())


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 90), cut=st.integers(1, 89), step0=st.sampled_from([0, 5, 2 ** 32 - 3]),
       rseed=st.integers(0, 2 ** 32 - 1))
def test_fp16_state_does_not_depend_on_the_shard_a_drone_is_in(seed, n, cut, step0, rseed):
    """fp16 storage: the stochastic rounding of a drone is keyed by its GLOBAL id (and the step index, and the buffer's
    rounding seed), so a population stepped as one batch and as two shards with drone_id_offset = the shard's first id
    ends in the same half-precision bits - odd cuts included, where the thrust halves of the two neighbours of the cut
    no longer share a word - and across the 2^32 step boundary."""
    cut = 1 + (cut - 1) % (n - 1)
    rng = np.random.default_rng(seed)
    steps = 12
    acts = rng.uniform(-1, 1, (steps, n, 4)).astype(np.float32)
    whole_pos, whole_sh = lane_model.split_half(lane_model.initial_state(P, n))
    lane_model.run_h(P, whole_pos, whole_sh, acts, seed0=rseed, step0=step0)
    whole = lane_model.join_half(whole_pos, whole_sh)
    for lo, hi in ((0, cut), (cut, n)):
        m = hi - lo
        pos, sh = lane_model.split_half(lane_model.initial_state(P, m))
        lane_model.run_h(P, pos, sh, np.ascontiguousarray(acts[:, lo:hi]), seed0=rseed, step0=step0, drone_id_offset=lo)
        part = lane_model.join_half(pos, sh)
        assert np.array_equal(part[:, :m].view(np.uint32), whole[:, lo:hi].view(np.uint32)), (lo, hi)


@settings(max_examples=60, deadline=None)
@given(q=st.lists(st.floats(-1, 1, allow_nan=False), min_size=4, max_size=4), tiny=st.floats(0, 1e-7))
def test_rotation_helpers_round_trip_for_any_attitude(q, tiny):
    """fpyv_amd.env.matrix_to_quat / quat_to_matrix / matrix_to_euler_zyx / euler_zyx_matrix on arbitrary unit quaternions
    (half turns included - w within `tiny` of zero - where the reference's own matrix -> quaternion formula divides by ~0):
    the matrix survives both round trips to 1e-12, the quaternion comes back with w >= 0 and unit norm."""
    import torch
    from fpyv_amd.env import euler_zyx_matrix, matrix_to_euler_zyx, matrix_to_quat, quat_to_matrix
    v = np.array(q, dtype=np.float64)
    if np.linalg.norm(v[1:]) < 1e-3:
        v[1] = 1.0
    v[0] *= tiny if tiny > 0 and abs(v[0]) < 0.5 else 1.0          # sometimes: an (almost) exact half turn
    v /= np.linalg.norm(v)
    R = quat_to_matrix(torch.from_numpy(v)[None])
    assert float((R @ R.transpose(-1, -2) - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-14
    back = matrix_to_quat(R)
    assert float(back[0, 0]) >= 0 and abs(float(back.norm()) - 1) < 1e-14
    assert float((quat_to_matrix(back) - R).abs().max()) < 1e-12
    e = matrix_to_euler_zyx(R)
    if abs(float(R[0, 2, 0])) < 1 - 1e-9:                          # away from gimbal lock the Euler round trip is exact too
        assert float((euler_zyx_matrix(e) - R).abs().max()) < 1e-9


@settings(max_examples=150, deadline=None)
@given(q=st.lists(st.floats(-1, 1, allow_nan=False, width=32), min_size=4, max_size=4),
       v=st.lists(st.floats(-80, 80, allow_nan=False, width=32), min_size=3, max_size=3),
       seed=st.integers(0, 2 ** 32 - 1), drone=st.integers(0, 2 ** 32 - 1))
def test_fp16_storage_round_trip_for_any_attitude_and_velocity(q, v, seed, drone):
    """The fp16 state's eleven words (fpv_pack_half -> fpv_unpack_half) for ANY unit quaternion and velocity: the decoded
    attitude is a unit quaternion within the fixed-point grid of the input (as an attitude: q and -q are the same), its
    largest component is the reconstructed one, the decoded velocity is within one 15-bit-mantissa step."""
    qn = np.asarray(q, dtype=np.float64)
    if np.linalg.norm(qn) < 1e-3:
        qn = np.array([1.0, 0, 0, 0])
    qn = (qn / np.linalg.norm(qn)).astype(np.float32)
    st14 = np.zeros((14, 64), dtype=np.float32)
    st14[3:6, 0], st14[6:10, 0] = v, qn
    pos, sh = lane_model.split_half(st14, seed=seed, drone_id_offset=drone)
    back = lane_model.join_half(pos, sh)[:, 0].astype(np.float64)
    qb = back[6:10]
    assert abs(np.linalg.norm(qb) - 1) < 2e-6
    grid = 1.0 / 23168
    d = min(np.abs(qb - qn).max(), np.abs(qb + qn).max())
    # three components within one grid step each; the fourth follows from |q| = 1 with a >= 1/2 pivot: <= ~3.5 steps
    assert d <= 3.6 * grid, (qn, qb, d / grid)
    # the dropped (reconstructed) component is the largest of the INPUT and the stored sign makes it positive; where two components
    # tie within the grid (q = (0, 0, s, -s): the stored one saturates at 16383 / 23168, a hair above the reconstructed
    # sqrt(1 - stored^2)), either of them may be the decoded maximum - one of the near-maximal components is the positive one
    m = np.abs(qb).max()
    assert m >= 0.5 - 1e-6 and any(qb[i] > 0 for i in range(4) if abs(qb[i]) >= m - 2.0 * grid), (qn, qb)
    vv = np.asarray(v, dtype=np.float32)
    # (spacing of |v|: numpy's spacing of a NEGATIVE power of two is the step of the binade below it)
    step = np.maximum(np.spacing(np.abs(vv).astype(np.float16)).astype(np.float64) / 32, 2.0 ** -24 * 1.01)
    assert np.all(np.abs(back[3:6] - vv) <= step * 1.0001 + 1e-12), (vv, back[3:6])


@settings(max_examples=150, deadline=None)
@given(pair=st.sampled_from([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]), sa=st.sampled_from([-1.0, 1.0]), sb=st.sampled_from([-1.0, 1.0]),
       scale=st.floats(1 - 1e-3, 1 + 1e-3), eps=st.floats(0, 1e-4), seed=st.integers(0, 2 ** 32 - 1), drone=st.integers(0, 2 ** 32 - 1))
def test_fp16_quaternion_fields_saturate_for_tied_components_off_the_unit_sphere(pair, sa, sb, scale, eps, seed, drone):
    """Two (nearly) tied largest components at +-1/sqrt(2) of a quaternion whose norm is off by up to 1e-3 (a user-written
    state, fp32 drift): the stored 15-bit field would reach 16384 and read back as -16384 - a sign flip.  The field
    saturates (fpv_q3_field): the decoded components keep the input's signs (as an attitude: up to q -> -q)."""
    qn = np.zeros(4)
    qn[pair[0]], qn[pair[1]] = sa * np.sqrt(0.5) * scale, sb * np.sqrt(0.5) * (scale - eps)
    qn = qn.astype(np.float32)
    st14 = np.zeros((14, 64), dtype=np.float32)
    st14[6:10, 0] = qn
    pos, sh = lane_model.split_half(st14, seed=seed, drone_id_offset=drone)
    qb = lane_model.join_half(pos, sh)[6:10, 0].astype(np.float64)
    if qb[pair[0]] * qn[pair[0]] < 0:
        qb = -qb
    assert abs(np.linalg.norm(qb) - 1) < 2e-6
    assert np.abs(qb - qn).max() < 2e-3 + 3.6 / 23168, (qn, qb)
    assert qb[pair[0]] * qn[pair[0]] > 0.49 and qb[pair[1]] * qn[pair[1]] > 0.49, (qn, qb)


@settings(max_examples=200, deadline=None)
@given(a=st.integers(0, 2 ** 31 - 1), b=st.integers(0, 2 ** 31 - 1), sign=st.integers(0, 1))
def test_inverse_normal_is_monotone_and_symmetric_for_any_pair_of_words(a, b, sign):
    """fpv_normal_from_word: a larger tail probability never gives a larger |z| (beyond the table's own 2.2e-6), the sign
    bit only flips the sign, and the value is the exact inverse CDF to 3e-6."""
    from oracle import philox
    lo, hi = sorted((a, b))
    w = np.array([lo, hi], dtype=np.uint32) | np.uint32(sign << 31)
    z = lane_model.normal_from_words(w).astype(np.float64)
    assert abs(z[1]) <= abs(z[0]) + 5e-6
    assert np.all(np.signbit(z) == bool(sign)) or np.any(z == 0)
    assert np.abs(z - philox.normal_from_words(w)).max() < 3e-6
