#!/usr/bin/env python3
"""Print VGPR / SGPR / occupancy / scratch of every gfx950 kernel in fpyv_amd/csrc/fpv_hip.hip
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed)."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_fpv_res.so", SRC]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            rows.append({"name": m.group(1)})
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and rows:
                rows[-1][key] = int(m.group(1))
    names = "\n".join(r["name"] for r in rows)
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'kernel':100s} vgpr sgpr occ scratch lds")
    for r, d in zip(rows, dem):
        d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
        if flt in d:
            print(f"{d[:100]:100s} {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('occ', -1):3d} {r.get('scratch', -1):7d} {r.get('lds', -1)}")


if __name__ == "__main__":
    main()
