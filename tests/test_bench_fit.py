"""bench.py's launch-time fit (roofline.launch_time_fit) says when its own points cannot be trusted: the driver's round-4
line printed a 12.8 us floor at 129 % of the HBM peak from a 2^19-drone leg that was not on the line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = 133          # algorithmic bytes per env-step of the fp32 step kernel
N = 1 << 20


def pts(us):
    return [(n, B * n, t) for n, t in zip((N // 2, N, 2 * N), us)]


def test_points_on_a_line_give_a_valid_fit():
    # the builder's round-4 measurement (profiles/archive/r04_bench_n1_step.json)
    f = bench.fit_launch_time(pts([13.124, 22.291, 40.751]))
    assert f["valid"] and f["invalid_reason"] is None
    assert 3.5 < f["floor_us"] < 4.5 and f["max_residual_us"] < 0.1
    assert 0.9 < f["streaming_frac_of_peak"] < 1.0 and abs(f["streaming_GBs"] / 8000.0 - f["streaming_frac_of_peak"]) < 1e-12
    assert f["drones"] == [N // 2, N, 2 * N]


def test_the_drivers_round4_points_are_flagged_invalid():
    # BENCH_r04.json: the 2^19 leg took as long as the 2^20 one
    f = bench.fit_launch_time(pts([22.004, 22.700, 41.072]))
    assert f["valid"] is False
    assert "residual" in f["invalid_reason"] and "above the 8000 GB/s HBM peak" in f["invalid_reason"]
    assert f["max_residual_us"] > 0.5 and f["streaming_frac_of_peak"] > 1.0


def test_a_flat_or_falling_line_and_a_negative_floor_are_invalid():
    assert bench.fit_launch_time(pts([20.0, 20.0, 20.0]))["valid"] is False          # no slope: "infinite" streaming rate
    assert bench.fit_launch_time(pts([30.0, 25.0, 20.0]))["valid"] is False
    f = bench.fit_launch_time(pts([5.0, 20.0, 50.0]))                                  # floor = -10 us
    assert f["valid"] is False and "not positive" in f["invalid_reason"]
    assert bench.fit_launch_time(pts([13.1, float("nan"), 40.7]))["valid"] is False


def test_median_helper():
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0]) == 2.5 and bench.median([7.0]) == 7.0


def test_roofline_frac_follows_value():
    """one line, one answer.  `roofline.frac` is `value` x algorithmic bytes / peak (the wall clock of the timed
    region); the HIP-event figure is `frac_events`.  Held on the helper and on every recorded headline line of this round."""
    import glob
    import json
    import os
    r = bench.hbm_roofline(49.71e9, 133, 1 << 20, 1.0, 20.08e-6)          # the driver's round-5 line: 49.71 G env-steps/s, 20.08 us per launch by events
    assert abs(r["frac"] - 49.71e9 * 133 / 8e12) < 1e-12 and abs(r["frac"] - 0.8264) < 1e-3
    assert abs(r["frac_events"] - 133 * (1 << 20) / 20.08e-6 / 8e12) < 1e-12 and abs(r["frac_events"] - 0.868) < 1e-3
    assert abs(r["achieved"] - r["frac"] * 8000.0) < 1e-9 and abs(r["achieved_events"] - r["frac_events"] * 8000.0) < 1e-9
    lines = sorted(glob.glob(os.path.join(bench.REPO, "profiles", "r06_bench_n1_step*.json")) + glob.glob(os.path.join(bench.REPO, "profiles", "r06_bench_n1_fp16.json")))
    for p in lines:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        ro = d["roofline"]
        assert abs(ro["frac"] - d["value"] / d["n_gpus"] * ro["algorithmic_bytes_per_env_step"] / 8e12) < 1e-9, p
        assert abs(ro["frac_events"] - ro["algorithmic_bytes_per_env_step"] * d["config"]["drones_per_gpu"] * d["config"]["steps_per_launch"] / (ro["avg_launch_us"] * 1e-6) / 8e12) < 1e-9, p
