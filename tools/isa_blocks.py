#!/usr/bin/env python3
"""Per-basic-block instruction statistics of one gfx950 kernel (hipcc -S --cuda-device-only; no GPU needed).

    python tools/isa_blocks.py <mangled-name-substring> [-D...]       # e.g.  rollout_kernelILb0ELb0ELb0ELb1E

Prints VALU / SALU / memory / branch counts per block and marks the blocks the compiler tagged as loops - the quick
way to see what a kernel's hot loop really issues (and whether spill traffic, flat or scratch accesses crept in)."""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from __graft_entry__ import HIPCC_FLAGS
    pat = next(a for a in sys.argv[1:] if not a.startswith("-"))
    extra = [a for a in sys.argv[1:] if a.startswith("-") and a != "--dump"]
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = "/tmp/_fpv_isa.s"
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", out,
                    os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    names = [m.group(1) for m in re.finditer(r"^(_Z\S+):\s*; @", s, re.M) if pat in m.group(1)]
    if not names:
        raise SystemExit(f"no kernel matches {pat!r}")
    name = names[0]
    a = s.index(name + ":")
    b = s.index(".amdhsa_kernel " + name)
    body = s[a:b]
    if "--dump" in sys.argv:
        print(body)
        return
    cur, stats = "entry", collections.OrderedDict()
    stats[cur] = collections.Counter()
    for ln in body.splitlines():
        m = re.match(r"(\.LBB\d+_\d+):(.*)", ln)
        if m:
            cur = m.group(1) + (" [loop]" if "Loop" in m.group(2) else "")
            stats[cur] = collections.Counter()
        elif ln.startswith("\t") and ln.strip() and not ln.startswith(("\t;", "\t.")):
            op = ln.split()[0]
            kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                    "mem" if op.startswith(("global_", "flat_", "buffer_", "scratch_", "ds_")) else "other")
            stats[cur][kind] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                stats[cur]["branch"] += 1
            if op.startswith(("v_readlane", "v_writelane")):
                stats[cur]["sgpr-spill-traffic"] += 1
            if op.startswith(("flat_", "scratch_")):
                stats[cur]["FLAT/SCRATCH"] += 1
            if op in ("v_sqrt_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32"):
                stats[cur]["quarter-rate"] += 1
    print(name)
    for k, v in stats.items():
        if sum(v.values()):
            print(f"  {k:22s} " + "  ".join(f"{a}={b}" for a, b in sorted(v.items())))


if __name__ == "__main__":
    main()
