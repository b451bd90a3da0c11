#!/usr/bin/env python3
"""How much do the kernel chains of the split-phase API overlap?  Reads a rocprofv3 kernel trace (CSV) of
`bench.py --partitions 2` and reports, for the step kernels of the timed part: kernels per queue, mean duration, the share
of the time during which TWO step kernels are running, and the time per full-population step.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_p2 -- python3 bench.py --partitions 2 --steps 2000 --warmup 200 --no-cpu-baseline --sustained-steps 0
    python3 tools/trace_overlap.py gpurun_out/trace_p2 [--tag r04]
"""
import argparse
import csv
import glob
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--tag", default=None)
ap.add_argument("--grid", default=None, help="only kernels of this Grid_Size (e.g. 524288 = the half-population kernels of two partitions)")
ap.add_argument("--kernels-per-step", type=int, default=2)
a = ap.parse_args()
f = max(glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "fpv_drone_step_kernel" in r["Kernel_Name"]]
if a.grid:
    rows = [r for r in rows if r.get("Grid_Size", r.get("Grid_Size_X")) == a.grid]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the steady part: the second half of the run (warm-up, preheat and ramp dropped)
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), +1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t_prev, depth, busy = ev[0][0], 0, {0: 0, 1: 0, 2: 0, 3: 0}
for t, d in ev:
    busy[min(depth, 3)] += t - t_prev
    t_prev, depth = t, depth + d
span = ev[-1][0] - ev[0][0]
queues = sorted({r.get("Queue_Id", "?") for r in rows})
grids = sorted({r.get("Grid_Size", r.get("Grid_Size_X", "?")) for r in rows})
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
per_q = {q: sum(1 for r in rows if r.get("Queue_Id", "?") == q) for q in queues}
lines = [f"# split phase under rocprofv3 --kernel-trace: {len(rows)} step kernels of the steady part, queues {per_q}, grid sizes {grids}",
         f"mean kernel duration {sum(dur) / len(dur) / 1e3:.2f} us",
         f"time with 0 / 1 / 2 step kernels running: {100 * busy[0] / span:.1f} % / {100 * busy[1] / span:.1f} % / {100 * (busy[2] + busy[3]) / span:.1f} %",
         f"time per full-population step ({a.kernels_per_step} kernel(s)): {span / (len(rows) / a.kernels_per_step) / 1e3:.2f} us"]
print("\n".join(lines))
if a.tag:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    open(os.path.join(here, "profiles", f"{a.tag}_partitions_overlap.txt"), "w").write("\n".join(lines) + "\n")
