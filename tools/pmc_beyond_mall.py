#!/usr/bin/env python3
"""Counter workload and summary for the step kernel BEYOND the Infinity Cache (2^23 drones, 470 MB of state) next to the
float4 copy of the same byte count (fpv_diag_stream_copy_wide), VERDICT r4 #4.

  workload (one rocprofv3 pass per counter group, the program directly after `--`):
      rocprofv3 --kernel-trace --pmc <4 counters> --output-format csv -d gpurun_out/r5_pmcb/<group> -- python3 tools/pmc_beyond_mall.py
  summary (reads every group directory, writes profiles/<round>_beyond_mall_counters.md):
      python3 tools/pmc_beyond_mall.py --summarise gpurun_out/r5_pmcb --round r05

GROUPS lists the passes: translation (UTCL1), the L2's memory-side requests and their stalls, L2 hit/miss, request latency."""
import argparse
import collections
import csv
import glob
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

GROUPS = {
    "utcl1_a": "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum",
    "utcl1_b": "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum",
    "ea_rd": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum",
    "ea_wr": "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
    "l2": "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum",
    "l2_b": "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum",
    "tcp_rd": "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum",
    "tcp_wr": "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum",
    "size": "FETCH_SIZE GRBM_GUI_ACTIVE",
    "wsize": "WRITE_SIZE",
}
KERNELS = (("step", "fpv_drone_step_kernel"), ("copy16", "fpv_diag_copy4_kernel"), ("copy4", "fpv_diag_copy_kernel"))


def workload(a):
    import torch
    from fpyv_amd import _lib, load_params, sticks
    from fpyv_amd.env import DroneBatch
    dev = torch.device("cuda:0")
    env = DroneBatch(load_params(fps=1000, ceiling=100.0), a.n, device=dev, auto_reset=True, with_accel=False)
    env.reset()
    acts = sticks.ema_noise_device(a.ring, a.n, dev, seed=99)
    for _ in range(a.launches // a.ring):
        env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    cf = (env.algorithmic_bytes() * a.n // 8) // 1024 * 1024          # read + write = one step launch's bytes
    src = torch.randn(cf, device=dev)
    dst = torch.empty_like(src)
    L = _lib.lib()
    for _ in range(a.launches // 2):
        _lib.check(L.fpv_diag_stream_copy_wide(dst.data_ptr(), src.data_ptr(), cf, None))
    for _ in range(a.launches // 4):
        _lib.check(L.fpv_diag_stream_copy(dst.data_ptr(), src.data_ptr(), cf, None))
    torch.cuda.synchronize()
    print("workload done", a.n, a.launches, cf, flush=True)


def read_group(d):
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        short = next((s for s, pat in KERNELS if pat in r["Kernel_Name"]), None)
        if short:
            agg[(short, r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return agg


def summarise(a):
    rows, durs = collections.OrderedDict(), collections.defaultdict(list)
    for g in GROUPS:
        agg = read_group(os.path.join(a.summarise, g))
        for (k, c), v in agg.items():
            v = v[len(v) // 4:]                               # the first launches warm up
            rows.setdefault(c, {})[k] = sum(x[0] for x in v) / len(v)
            durs[k].append(sum(x[1] for x in v) / len(v))
    n = a.n
    out = [f"# Counters of the step kernel beyond the Infinity Cache ({n} drones) next to the copies of the same byte count", "",
           "`tools/pmc_beyond_mall.py`: one `rocprofv3 --kernel-trace --pmc` pass per group (the program directly after `--`), per-launch "
           "means over the later three quarters of the launches.  `step` = `fpv_drone_step_kernel<0,0,0,0>` (14 + 14 dword row streams, a 16-byte "
           "action row, reward, done: 133 B per drone), `copy16` = `fpv_diag_stream_copy_wide` (float4 per lane), `copy4` = `fpv_diag_stream_copy` "
           "(one dword per lane); all three move the same bytes per launch (read + write = 133 x n).", "",
           "| counter | step | copy16 | copy4 | step / copy16 |", "|---|---|---|---|---|"]
    for c, d in rows.items():
        s, c16, c4 = d.get("step"), d.get("copy16"), d.get("copy4")
        fmt = lambda x: "-" if x is None else f"{x:.4g}"      # noqa: E731
        out.append(f"| {c} | {fmt(s)} | {fmt(c16)} | {fmt(c4)} | {fmt(s / c16 if s is not None and c16 else None)} |")
    out += ["", "Mean kernel duration inside the counter passes (ns; counters slow a launch a little): "
            + ", ".join(f"{k} {sum(v) / len(v):.0f}" for k, v in durs.items()), ""]
    bytes_launch = 133 * n
    if "FETCH_SIZE" in rows and "WRITE_SIZE" in rows:
        f, w = rows["FETCH_SIZE"], rows["WRITE_SIZE"]
        out += ["HBM traffic (FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads half of a coalesced streaming read - guide, HBM section - "
                "so the read side is calibrated on the copy of the same access width in the same pass):", ""]
        for k, cal in (("step", "copy4"), ("copy16", "copy16"), ("copy4", "copy4")):
            if k in f and cal in f and k in w:
                scale = (bytes_launch / 2 / 1024) / f[cal]
                rd, wr = f[k] * scale * 1024, w[k] * 1024
                out.append(f"* {k}: read {rd / 1e6:.1f} MB (FETCH_SIZE x {scale:.3f}), written {wr / 1e6:.1f} MB, total {(rd + wr) / 1e6:.1f} MB "
                           f"= {(rd + wr) / bytes_launch:.4f} x the algorithmic {bytes_launch / 1e6:.1f} MB")
    path = os.path.join(REPO, "profiles", f"{a.round}_beyond_mall_counters.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 23)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--summarise", default=None)
    ap.add_argument("--round", default="r05")
    ap.add_argument("--print-groups", action="store_true")
    a = ap.parse_args()
    if a.print_groups:
        for g, c in GROUPS.items():
            print(g, c)
    elif a.summarise:
        summarise(a)
    else:
        workload(a)
