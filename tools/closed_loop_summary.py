#!/usr/bin/env python3
"""profiles/<round>_closed_loop.md from what tools/gpu/closed_loop.sh left under gpurun_out/<round>/: the in-process A/B of the
rotation inside the closed loop (closed_loop_ab.json) and the rocprofv3 kernel statistics of examples/closed_loop_policy.py
with the rotation automatic and off (cl_kt_<drones>_rot<-1|0>_kernel_stats.csv, cl_kt_mlp_kernel_stats.csv).

    python tools/closed_loop_summary.py r06 > profiles/r06_closed_loop.md
"""
import csv
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, "gpurun_out", R)
HBM = 8000.0


def short(name):
    if name.startswith("Cijk"):
        mt = name.split("_MT")[1].split("_")[0] if "_MT" in name else "?"
        return f"rocBLAS/Tensile GEMM `Cijk_Ailk_Bljk_S..._MT{mt}_MI16x16x1` (the policy's `W @ obs`)"
    if "fpv_drone_step_kernel" in name:
        return "`fpv_drone_step_kernel<0,0,0,0>` (the stepper)"
    if "tanh_kernel" in name:
        return "torch `tanh` (elementwise, [4, N])"
    if "launch_clamp" in name or "relu" in name.lower() or "clamp" in name:
        return "torch `relu` (elementwise, [64, N])"
    return "`" + name.split("(")[0][-60:] + "`"


def stats(tag, min_calls):
    p = os.path.join(O, f"{tag}_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(p)) if int(r["Calls"]) >= min_calls]
    return [(short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows]


ab = json.load(open(os.path.join(O, "closed_loop_ab.json")))
print(f"# {R}: the closed loop action -> step -> action (`/root/reference/src/core/simulator.py:83-156`) on one MI355X\n")
print("Policy: `tanh(W[4,13] @ obs[13,N])` on the LIVE state view (`obs = state[:13, :n]`, row stride `ld`), sticks consumed in place as SoA `[4, N]`"
      " (`fpv_buffers_t.action_ld`): no transpose, no copy.  Sources: `tools/closed_loop_ab.py` (one process, interleaved rounds, HIP events; median of"
      f" {ab['rounds']}) and `rocprofv3 --kernel-trace --stats -- python3 examples/closed_loop_policy.py --partitions 1 --rotation <-1|0>`"
      " (`tools/gpu/closed_loop.sh`).\n")
print("## 1. Does the rotation of the traversal survive a policy kernel between steps?\n")
print("| drones | rotation (auto) | closed loop, auto | closed loop, plain order | gain | step-only chain, auto | step-only, plain | gain |")
print("|---|---|---|---|---|---|---|---|")
for s in ab["sizes"]:
    m = s["us_per_step"]
    print(f"| {s['drones']} | {s['rotation_auto_drones']} | {m['closed_auto']['median']:.1f} us | {m['closed_plain']['median']:.1f} us | "
          f"{100 * (s['closed_loop_rotation_gain'] - 1):+.1f} % | {m['step_auto']['median']:.2f} us | {m['step_plain']['median']:.2f} us | {100 * (s['step_only_rotation_gain'] - 1):+.1f} % |")
print("\n(`DroneBatch` defaults of the A/B: accel rows on; they are written with a streaming hint and do not change the automatic rotation.)\n")
print("The step kernel alone, inside the loop (kernel trace, average over the timed + warm-up launches):\n")
print("| drones | step kernel, rotation auto | step kernel, plain order | GEMM, auto | GEMM, plain |")
print("|---|---|---|---|---|")
per = {}
for n in (1048576, 8388608):
    row = {}
    for rot in (-1, 0):
        try:
            st = stats(f"cl_kt_{n}_rot{rot}", 100)
        except FileNotFoundError:
            continue
        row[rot] = {("step" if "stepper" in k else "gemm" if "GEMM" in k else "tanh" if "tanh" in k else k): us for k, _, us in st}
        per[(n, rot)] = st
    if len(row) == 2:
        print(f"| {n} | {row[-1]['step']:.2f} us | {row[0]['step']:.2f} us | {row[-1]['gemm']:.2f} us | {row[0]['gemm']:.2f} us |")
print("""
**Answer: no - and it does not need to.**  With a pass of another kernel over the state between two steps, what the eight L2s and
the Infinity Cache hold when a step starts is what THAT kernel touched last, not what the previous step wrote last: the rotated and
the plain order run the step kernel within 1 % of each other, and the whole loop within 1.5 %.  The rotation's gain (claims 1, 8 of
DESIGN.md) is a property of STEP-ONLY chains - `fpv_rollout`, in-kernel stick noise, precomputed sticks: the shape of BASELINE
configs[2], whose sticks are a noise profile, not a policy - and the bench line's `cache_note` says so.  In the closed loop the
step kernel still finds most of its state on the chip (the GEMM has just read 13 of its 14 rows): at 2^20 drones it runs at the
step-only chain's speed, at 2^23 drones FASTER than the step-only plain order.  The rotation is left on in both cases: it costs nothing.
""")
print("## 2. Which kernels are the loop?\n")
for (n, rot), st in sorted(per.items()):
    if rot != -1:
        continue
    tot = sum(us for _, _, us in st)
    print(f"{n} drones, linear policy, per step ({tot:.1f} us of kernel time):\n")
    print("| kernel | calls | average | share | algorithmic bytes | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for k, calls, us in st:
        b = 133 * n if "stepper" in k else (13 * 4 + 16) * n if "GEMM" in k else 32 * n if "tanh" in k else None
        print(f"| {k} | {calls} | {us:.2f} us | {100 * us / tot:.1f} % | {b / 1e6:.1f} MB | {b / us / 1e3 / HBM:.3f} |" if b else f"| {k} | {calls} | {us:.2f} us | {100 * us / tot:.1f} % | | |")
    print()
try:
    st = stats("cl_kt_mlp", 100)
    tot = sum(us for _, _, us in st)
    print(f"1048576 drones, MLP 13-64-4 policy, per step ({tot:.1f} us of kernel time):\n")
    print("| kernel | calls | average | share |")
    print("|---|---|---|---|")
    for k, calls, us in st:
        print(f"| {k} | {calls} | {us:.2f} us | {100 * us / tot:.1f} % |")
    print()
except FileNotFoundError:
    pass
print("## 3. Is it the stepper's layout that the GEMM pays for?\n")
print("The same product on the live state view (row stride `ld` = `fpv_recommended_ld(n)`: padded), on a contiguous `[13, N]` copy, into a"
      " preallocated output, and transposed (`obs^T @ W^T -> [N, 4]` rows, the stepper's other action layout); microseconds per call, median:\n")
print("| drones | ld | on the state view | contiguous copy | view, preallocated out | transposed, rows out | transposed, contiguous | tanh [4,N] | GEMM bytes / time |")
print("|---|---|---|---|---|---|---|---|---|")
for s in ab["sizes"]:
    p = s["policy_us_per_call"]
    g = p["gemm_on_state_view"]["median"]
    print(f"| {s['drones']} | {s['ld']} | {g:.2f} | {p['gemm_on_contiguous_copy']['median']:.2f} | {p['gemm_on_state_view_out_preallocated']['median']:.2f} | "
          f"{p['gemm_transposed_rows_out']['median']:.2f} | {p['gemm_transposed_rows_out_contiguous']['median']:.2f} | {p['tanh_4xN']['median']:.2f} | "
          f"{s['policy_algorithmic_bytes']['gemm'] / g / 1e3:.0f} GB/s = {s['policy_algorithmic_bytes']['gemm'] / g / 1e3 / HBM:.3f} of peak |")
print("""
**Answer: no.**  The padded row stride costs the GEMM under 1 % against a contiguous copy, the SoA `[4, N]` output as much as the
`[N, 4]` one.  The 4 x 13 x N product is 68 B per drone of traffic and the library's skinny-GEMM kernel (a 256 x 16 MFMA macro-tile
with 4 useful rows of 256) moves them at about an eighth of the HBM peak at either size: the loop's 70 % is the library kernel's shape,
not a layout the stepper chose.  Nothing on the stepper's side to change; a policy kernel is outside SURVEY.md section 8.  Closed.""")
