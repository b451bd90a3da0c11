#!/usr/bin/env python3
"""Print VGPR / SGPR / spills / occupancy / scratch of every gfx950 kernel in fpyv_amd/csrc/fpv_hip.hip
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).

    python tools/kernel_resources.py [name-filter] [-D...]      # extra -D flags go to hipcc

The compiler prints `TotalSGPRs:` (the round-2 version of this tool looked for ` SGPRs:`, matched only the
`SGPRs Spill:` line by accident of ordering and showed -1 everywhere) and, separately, `SGPRs Spill:` /
`VGPRs Spill:`: a spilled SGPR lives in a VGPR lane and costs a v_readlane / v_writelane per use.
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")
KEYS = (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("sgpr", r"remark:\s+TotalSGPRs: (\d+)"), ("sspill", r"remark:\s+SGPRs Spill: (\d+)"),
        ("vspill", r"remark:\s+VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))


def collect(extra_flags=()):
    sys.path.insert(0, REPO)
    from __graft_entry__ import HIPCC_FLAGS
    cmd = ["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_fpv_res.so", SRC]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            rows.append({"name": m.group(1)})
            continue
        for key, pat in KEYS:
            m = re.search(pat, ln)
            if m and rows:
                rows[-1][key] = int(m.group(1))
    names = "\n".join(r["name"] for r in rows)
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, dem):
        r["kernel"] = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    return rows


def main():
    flt = next((a for a in sys.argv[1:] if not a.startswith("-")), "")
    rows = collect([a for a in sys.argv[1:] if a.startswith("-")])
    print(f"{'kernel':86s} vgpr sgpr s-spill v-spill occ scratch lds")
    for r in rows:
        if flt in r["kernel"]:
            print(f"{r['kernel'][:86]:86s} {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('sspill', -1):7d} {r.get('vspill', -1):7d} "
                  f"{r.get('occ', -1):3d} {r.get('scratch', -1):7d} {r.get('lds', -1)}")
    print(f"# {len(rows)} kernels")


if __name__ == "__main__":
    main()
