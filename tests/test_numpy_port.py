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
