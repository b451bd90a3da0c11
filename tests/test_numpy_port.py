"""The NumPy-vectorised restatement (oracle/numpy_port.py, SURVEY 8(d)(iii): the idiomatic-Python CPU baseline) against
the C oracle and against the reference captures - it is a second, independently written checker, so agreement to
rounding level pins both."""
import numpy as np
import pytest

from conftest import load_golden, params_for_golden
from oracle import numpy_port, oracle


@pytest.mark.parametrize("name", ["g2_sin_4096", "g3_ema_noise", "g4_saturated", "g5_attitude_wind", "g6_ground"])
def test_numpy_port_equals_c_oracle_and_reference_capture(params_1k, name):
    g = load_golden(name)
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    S = numpy_port.initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    ref = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    done_np = np.zeros((n, T), dtype=bool)
    for t in range(T):
        acc, done_np[:, t] = numpy_port.step(params_1k, S, acts[t], g["wind"])
    _, ref_acc, ref_done = oracle.drone_run(params_1k, ref, acts, wind=g["wind"])
    rows = numpy_port.as_oracle_rows(S)
    scale = np.maximum(np.abs(ref), 1.0)
    assert (np.abs(rows - ref) / scale).max() < 1e-11, (np.abs(rows - ref) / scale).max()
    assert np.abs(acc - ref_acc).max() < 1e-9 * max(1.0, np.abs(ref_acc).max())
    assert np.array_equal(done_np[:, -1].astype(np.uint8), ref_done)
    assert np.array_equal(done_np.astype(np.uint8), g["done"]), "the ground flag must flip on exactly the reference's steps"
    want = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1], g["prev_thrust"][:, -1:]], axis=1)
    assert (np.abs(rows - want) / np.maximum(np.abs(want), 1.0)).max() < 1e-11


@pytest.mark.parametrize("k", range(4))
def test_numpy_port_on_the_other_drone_types(k):
    """Capture G14 (every constructor parameter away from params.yaml): the independently written NumPy port lands on
    the reference's numbers too."""
    g = load_golden(f"g14_drone_type_{k}")
    p = params_for_golden(g)
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    S = numpy_port.initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    for t in range(T):
        numpy_port.step(p, S, acts[t], g["wind"])
    rows = numpy_port.as_oracle_rows(S)
    want = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1], g["prev_thrust"][:, -1:]], axis=1)
    assert (np.abs(rows - want) / np.maximum(np.abs(want), 1.0)).max() < 1e-11


@pytest.mark.parametrize("name", ["g2_sin_4096", "g3_ema_noise", "g4_saturated", "g5_attitude_wind", "g6_ground"])
def test_simd_across_drones_build_equals_the_scalar_oracle(params_1k, name):
    """oracle/fpv_oracle_simd.c (bench.py's cpu_baseline.simd_across_drones leg: SoA tiles, `omp simd` over drones, libmvec
    sin / cos) against the scalar C oracle at 1e-12 and against the reference capture: the same arithmetic as
    /root/reference/src/utils/components.py:220-248, including the ground flag's steps (G6)."""
    g = load_golden(name)
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    a = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    b = a.copy()
    p = params_1k
    _, acc_a, done_a = oracle.drone_run(p, a, acts, wind=g["wind"])
    _, acc_b, done_b = oracle.drone_run_simd(p, b, acts, wind=g["wind"], threads=2)
    scale = np.maximum(np.abs(a), 1.0)
    assert (np.abs(a - b) / scale).max() < 1e-12, (np.abs(a - b) / scale).max()
    assert np.abs(acc_a - acc_b).max() < 1e-10 * max(1.0, np.abs(acc_a).max()) and np.array_equal(done_a, done_b)
    want = np.concatenate([g["state"][:, -1], g["R"][:, -1].reshape(n, 9), g["prev_rates"][:, -1], g["prev_thrust"][:, -1:]], axis=1)
    assert (np.abs(b - want) / np.maximum(np.abs(want), 1.0)).max() < 1e-11
    # step by step: the done flag of every step, ragged sizes (a tile is 256 drones), a held action
    m = min(n, 3) if n < 300 else 300
    c, d = a[:m].copy(), a[:m].copy()
    for t in range(min(T, 40)):
        _, _, d1 = oracle.drone_run(p, c, acts[t, :m], steps=1, wind=g["wind"])
        _, _, d2 = oracle.drone_run_simd(p, d, acts[t, :m], steps=1, wind=g["wind"])
        assert np.array_equal(d1, d2)
    assert (np.abs(c - d) / np.maximum(np.abs(c), 1.0)).max() < 1e-12


def test_simd_across_drones_ground_contact_and_refusal(params_1k):
    """Drones dropped onto the ground plane: the spring force of components.py:198-214 and the crash flag through the
    mask arithmetic of the vector form; a general object list is refused, not approximated."""
    p = params_1k.replace(ground=True, init_position=np.array([0.0, 0.0, 0.05]), init_velocity=np.array([0.3, 0.0, -1.0]))
    n, T = 700, 400
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (T, n, 4)) * np.array([0.3, 0.3, 0.3, 1.0])
    a = oracle.drone_initial_state(n, p.init_position, p.init_velocity, p.init_orientation_deg)
    a[:, 2] += rng.uniform(0, 0.2, n)
    a[:, 5] = -rng.uniform(0.5, 9.0, n)            # slow ones bounce on the spring, fast ones go through the plane: done
    b = a.copy()
    dones = 0
    spring = False
    for t0 in range(0, T, 2):               # the flag is not latched (components.py:239-240): look at it every other step
        z_before = a[:, 2].copy()
        _, _, da = oracle.drone_run(p, a, acts[t0:t0 + 2])
        _, _, db = oracle.drone_run_simd(p, b, acts[t0:t0 + 2], threads=3)
        assert np.array_equal(da, db)
        dones += int(da.sum())
        spring |= bool(((z_before < 0.2) & (z_before > 0)).any())
    assert spring
    assert dones > 0, "some drones must have hit the ground"
    assert (np.abs(a - b) / np.maximum(np.abs(a), 1.0)).max() < 1e-12
    with pytest.raises(ValueError):
        oracle.drone_run_simd(params_1k.replace(objects=((1, 1.0, 0.0, 0.0, 0.2, 1.0),)), b, acts[:2])      # one Cylinder row
