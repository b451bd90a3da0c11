"""Order of the traversal, row stride, launch machinery: the rotation (bit-identical, the automatic rule), whole rounds of the XCDs,
results independent of ld, handle lifecycles, the stream probe's surface, bench.py's line.  Rules and bit identity only - what the
rotation is worth in TIME is tests/test_gpu_timing.py (marker gpu_timing, not part of the gate)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fpyv_amd import _lib, load_params, sticks
from gpu_helpers import DEV
from oracle import lane_model, oracle
from parity import REL_TOL, assert_parity, soa_vs_oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU: the stepper has no CPU path")]


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
    # the rule applies because the device is the one the cache model was measured on (fpv_create asked it: ABI 8); elsewhere the
    # automatic setting is the plain order (tests/test_device_guard.py drives that through a stand-in runtime)
    probe = DroneBatch(params_1k, 4096, device=DEV, with_accel=False)
    cm = probe.cache_model
    assert cm["matches"] and cm["arch"].startswith("gfx950") and cm["compute_units"] == 256 and cm["xcds"] == 8 and cm["reason"] is None, cm
    assert cm["l2_bytes_per_xcd"] in (0, 4 << 20) and cm["infinity_cache_bytes"] == MALL
    L = _lib.lib()
    for n in (4096, 1 << 19, 1_000_000, 1 << 20, 3 << 20):
        assert L.fpv_recommended_ld_device(n, 0) == L.fpv_recommended_ld(n)
    assert L.fpv_recommended_ld_device(1 << 20, 99) < 0 and b"device index" in L.fpv_last_error()
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
