#!/usr/bin/env python3
"""VALU / SALU instructions per wave of the step and k-step kernels from a counter-only rocprofv3 pass, written to
profiles/pmc_valu.json together with the hash of the kernel sources they were measured on (bench.py --api rollout
prices the k-step kernel against the vector-ALU issue peak with these counts, and only while the hash still matches).

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_valu -- \\
        python3 tools/kernel_sweep.py --fp16 --racer --fused --rounds 1 --launches 64 --ring 32
    python3 tools/pmc_valu.py gpurun_out/pmc_valu --steps-per-launch 32 --round r03
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps-per-launch", type=int, default=32)
    ap.add_argument("--drones", type=int, default=1 << 20)
    ap.add_argument("--round", default="r03")
    a = ap.parse_args()
    from bench import kernel_source_hash
    f = max(glob.glob(os.path.join(a.dir, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if "fpv_" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = {}
    for k, c in sorted(agg.items()):
        # one dispatch = one set of counter rows; take the per-dispatch medians (every dispatch of a kernel runs the same code path)
        w = sorted(c.get("SQ_WAVES", [0]))[len(c.get("SQ_WAVES", [0])) // 2]
        v = sorted(c.get("SQ_INSTS_VALU", [0]))[len(c.get("SQ_INSTS_VALU", [0])) // 2]
        s = sorted(c.get("SQ_INSTS_SALU", [0]))[len(c.get("SQ_INSTS_SALU", [0])) // 2]
        rows[k] = {"waves": w, "valu_per_wave": v / max(w, 1), "salu_per_wave": s / max(w, 1), "dispatches": len(c.get("SQ_WAVES", []))}
        print(f"{k[:84]:84s} waves {w:9.0f}  valu/wave {v / max(w, 1):9.1f}  salu/wave {s / max(w, 1):8.1f}  dispatches {rows[k]['dispatches']}")

    def pick(pred):
        for k, r in rows.items():
            if pred(k):
                return dict(kernel=k, steps_per_launch=a.steps_per_launch, valu_per_wave=r["valu_per_wave"], salu_per_wave=r["salu_per_wave"])
        return None

    fam = {"f32": pick(lambda k: k.startswith("fpv_drone_rollout_kernel<false, false, false")),
           "fp16": pick(lambda k: k.startswith("fpv_drone_rollout_h_kernel")),
           "racer": pick(lambda k: k.startswith("fpv_racer_rollout_kernel<true, false"))}
    out = {"source": f"profiles/{a.round}_pmc_valu_counts.log (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES, counters only; tools/pmc_valu.py)",
           "kernel_source_sha256_16": kernel_source_hash(), "library_sha256_16": __import__("bench").library_hash(), "drones": a.drones,
           "kernels": {k: v for k, v in fam.items() if v}, "all": rows}
    json.dump(out, open(os.path.join(REPO, "profiles", "pmc_valu.json"), "w"), indent=1)
    print("wrote profiles/pmc_valu.json for sources", out["kernel_source_sha256_16"])


if __name__ == "__main__":
    main()
