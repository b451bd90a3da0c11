"""Host-side parameter derivation against constants captured from the reference's Drone
constructor (/root/reference/src/utils/components.py:96-142) - tests/golden/params_golden.npz."""
import os

import numpy as np
import pytest

from conftest import load_golden
from fpyv_amd import read_motor_test_report
from fpyv_amd.params import ypr_to_quat, DEFAULT_PARAMS_PATH, params_from_dict

REF_CSV = "/root/reference/config/t_motos_f80_motor_test.csv"
REF_YAML = "/root/reference/config/params.yaml"


def test_thrust_curve_coefficients(params_1k):
    g = load_golden("params_golden")
    np.testing.assert_allclose(params_1k.thrust_poly, g["thrust_poly"], rtol=1e-12)
    np.testing.assert_allclose(params_1k.inverse_thrust_poly, g["inverse_thrust_poly"], rtol=1e-11)
    # SURVEY App. A anchors
    np.testing.assert_allclose(params_1k.thrust_poly, [-3.5693188139684359e-05, 9.0016725594999486e-03,
                                                       2.7025509193863934e-01, -4.6756286242420328e-02], rtol=1e-10)
    assert abs(params_1k.min_throttle_in_force - float(g["min_throttle_in_force"])) < 1e-12
    assert abs(params_1k.max_throttle_in_force - float(g["max_throttle_in_force"])) < 1e-11
    np.testing.assert_allclose(params_1k.thrust_from_stick(g["stick_samples"]), g["thrust_samples"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(params_1k.stick_from_thrust(np.array([0.0, 5.0, 31.5, 60.0, 90.0])),
                               g["thrust2throttle_samples"], rtol=1e-10, atol=1e-12)


def test_derived_constants(params_1k, params_60):
    g = load_golden("params_golden")
    assert params_1k.dt == float(g["dt"]) == 1 / 1000
    assert params_60.dt == 1 / 60
    assert params_1k.mass == float(g["mass"]) and params_1k.gravity == float(g["gravity"])
    assert params_1k.max_rates == float(g["max_rates"])
    np.testing.assert_array_equal(params_1k.drag_coefficients, g["drag_coef"])
    np.testing.assert_allclose(params_1k.cross_section_areas, g["cross_section_areas"], rtol=1e-15)
    np.testing.assert_allclose(params_1k.motor_xy, g["motors_relative_position"][:, :2], rtol=1e-15, atol=1e-18)
    assert np.all(g["motors_relative_position"][:, 2] == 0)
    assert params_1k.rates_transition_rate == float(g["rates_transition_rate"])
    assert params_1k.thrust_transition_rate == float(g["thrust_transition_rate"])


def test_packaged_table_matches_reference_blocks():
    g = load_golden("params_golden")
    blocks = read_motor_test_report(os.path.join(os.path.dirname(DEFAULT_PARAMS_PATH), "f80_thrust_table.csv"))
    assert len(blocks) == int(g["n_blocks"]) == 5
    for b, thr, grams in zip(blocks, g["block_throttle"], g["block_thrust_g"]):
        np.testing.assert_array_equal(b["throttle"], thr)
        np.testing.assert_array_equal(b["thrust"], grams)


def test_raw_vendor_format(tmp_path):
    """'%' throttle cells, decimal commas, header row, label cells - flight_time_calculator.py:23-39."""
    raw = tmp_path / "raw.csv"
    raw.write_text('Type,Propeller,Throttle,Thrust (g),V,A,RPM,Power (W),Eff,Temp\n'
                   ',,50%,100.5,1,1,1,1,1,\n'
                   'X,Y,100%,"200,25",1,"1,5",1,"3,5",1,\n'
                   ',,50%,50,1,1,1,1,1,\n'
                   ',,100%,80,1,1,1,1,1,\n')
    blocks = read_motor_test_report(str(raw))
    assert len(blocks) == 2
    np.testing.assert_array_equal(blocks[0]["thrust"], [100.5, 200.25])
    np.testing.assert_array_equal(blocks[1]["throttle"], [50, 100])


@pytest.mark.skipif(not os.path.isfile(REF_CSV), reason="reference mount not present (GPU box)")
def test_reference_files_load_unchanged():
    """A reference params.yaml + raw CSV give the same constants as the packaged defaults."""
    g = load_golden("params_golden")
    blocks = read_motor_test_report(REF_CSV)
    np.testing.assert_array_equal(np.stack([b["thrust"] for b in blocks]), g["block_thrust_g"])
    import yaml
    cfg = yaml.safe_load(open(REF_YAML, encoding="utf-8"))
    before = repr(cfg)
    p = params_from_dict(cfg, yaml_dir=os.path.dirname(REF_YAML), fps=1000)
    assert repr(cfg) == before, "the loader must not mutate the caller's dict (the reference ctor does)"
    np.testing.assert_allclose(p.thrust_poly, g["thrust_poly"], rtol=1e-12)
    np.testing.assert_array_equal(p.init_position, [0, 0, 10])


def test_ypr_quaternion():
    np.testing.assert_allclose(ypr_to_quat(0, 0, 0), [1, 0, 0, 0])
    from oracle import oracle
    for ypr in ([30, -20, 45], [170, 80, -100], [-90, 0, 180]):
        q = ypr_to_quat(*ypr)
        R = oracle.euler_zyx_matrix(*np.deg2rad(ypr))
        np.testing.assert_allclose(oracle.quat_to_matrix(q)[0], R, atol=1e-15)


def test_min_throttle_assert(params_1k):
    import yaml
    cfg = yaml.safe_load(open(DEFAULT_PARAMS_PATH))
    cfg["simulator"]["gravity"] = -9.81      # flips the sign of the bench thrust -> 5 % thrust < 0
    with pytest.raises(ValueError, match="minimum throttle"):
        params_from_dict(cfg)


def test_rotation_helpers_against_the_reference():
    """Capture G17: helper_functions' Euler -> matrix -> Euler and matrix <-> quaternion (helper_functions.py:39-80,
    :100-117) on 64 seeded attitudes; the build's torch helpers (fpyv_amd.env) run on CPU tensors here."""
    import torch
    from fpyv_amd.env import euler_zyx_matrix, matrix_to_euler_zyx, matrix_to_quat, quat_to_matrix
    g = load_golden("g17_rotation_helpers")
    ang = torch.from_numpy(g["euler_in"])
    R = euler_zyx_matrix(ang)
    np.testing.assert_allclose(R.numpy(), g["matrix"], atol=1e-14)
    np.testing.assert_allclose(matrix_to_euler_zyx(torch.from_numpy(g["matrix"])).numpy(), g["euler_out"], atol=1e-12)
    q = matrix_to_quat(torch.from_numpy(g["matrix"]))
    assert bool((q[:, 0] >= 0).all()) and float((q.norm(dim=1) - 1).abs().max()) < 1e-14
    np.testing.assert_allclose(q.numpy(), g["quat_wxyz"], atol=1e-9)              # the reference's own form loses digits near half turns
    np.testing.assert_allclose(quat_to_matrix(q).numpy(), g["matrix"], atol=1e-13)
    np.testing.assert_allclose(quat_to_matrix(torch.from_numpy(g["quat_wxyz"])).numpy(), g["matrix_from_quat"], atol=1e-13)
    # half-turn attitudes, where 1 + trace vanishes and the reference's formula divides by ~0: the build's stays exact
    half = torch.tensor([[[1.0, 0, 0], [0, -1, 0], [0, 0, -1]], [[-1.0, 0, 0], [0, 1, 0], [0, 0, -1]], [[-1.0, 0, 0], [0, -1, 0], [0, 0, 1]]],
                        dtype=torch.float64)
    qh = matrix_to_quat(half)
    np.testing.assert_allclose(quat_to_matrix(qh).numpy(), half.numpy(), atol=1e-15)
