#!/usr/bin/env python3
"""The ISA facts DESIGN.md quotes for the hot kernels, taken from a FRESH gfx950 disassembly (hipcc -S --cuda-device-only with
the shipped flags; no GPU needed): the load block as the compiler scheduled it, static instruction counts (loads, stores,
VALU, SALU, s_waitcnt, v_med3, LDS, scratch, MFMA) and - from -Rpass-analysis - VGPRs / SGPRs / spills / occupancy, for
the plain single-step kernel, the fp16-state single-step kernel and the plain k-step kernel.

    python tools/hot_kernel_isa.py                     # prints the report
    python tools/hot_kernel_isa.py --write r05         # also writes profiles/r05_hot_kernel_isa.txt

tests/test_isa_claims.py runs the same functions and holds DESIGN.md to the numbers.
"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SRC = os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")
HOT = collections.OrderedDict([
    ("plain single-step kernel fpv_drone_step_kernel<false,false,false,false>", "fpv_drone_step_kernelILb0ELb0ELb0ELb0E"),
    ("fp16-state single-step kernel fpv_drone_step_h_kernel", "fpv_drone_step_h_kernel"),
    ("plain k-step kernel fpv_drone_rollout_kernel<false,false,false,true>", "fpv_drone_rollout_kernelILb0ELb0ELb0ELb1E"),
])


def disassemble(path="/tmp/_fpv_claims.s"):
    """(assembly text, remark text) of the whole library at the shipped flags."""
    from __graft_entry__ import HIPCC_FLAGS
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", path, SRC],
                       capture_output=True, text=True, check=True)
    return open(path).read(), r.stderr


def kernel_bodies(asm):
    """{mangled name: instruction lines} for every kernel of the listing"""
    out = {}
    for m in re.finditer(r"^(_Z\S+):\s*; @", asm, re.M):
        name = m.group(1)
        a = m.end()
        b = asm.index(".amdhsa_kernel " + name, a)
        out[name] = [ln.strip() for ln in asm[a:b].splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", "."))]
    return out


def counts(lines):
    c = collections.Counter()
    for ln in lines:
        op = ln.split()[0]
        if op.startswith("global_load"):
            c["global_load"] += 1
            if re.match(r"global_load_dword v\d+, v\[\d+:\d+\], off", ln):
                c["global_load_dword_vaddr"] += 1               # 64-bit address in a VGPR pair
            if re.search(r", s\[\d+:\d+\]", ln):
                c["global_load_saddr"] += 1                     # SGPR base + 32-bit VGPR offset
        elif op.startswith("global_store"):
            c["global_store"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("scratch_", "buffer_")) or "flat_" in op:
            c["scratch_flat_buffer"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op == "s_waitcnt":
            c["s_waitcnt"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if op.startswith("v_lshl_add_u64"):
                c["v_lshl_add_u64"] += 1
            if op.startswith("v_med3_f32"):
                c["v_med3_f32"] += 1
            if op.startswith(("v_readlane", "v_writelane")):
                c["sgpr_spill_lane_ops"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def resources(remarks):
    rows, cur = {}, None
    for ln in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("sgpr", r"remark:\s+TotalSGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
                         ("vspill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def load_block(lines):
    """the instructions from the first vector load to the first wait for vector memory"""
    out, started = [], False
    for ln in lines:
        if ln.startswith("global_load"):
            started = True
        if started:
            out.append(ln)
            if ln.startswith("s_waitcnt vmcnt"):
                break
    return out


def report():
    asm, rem = disassemble()
    bodies, res = kernel_bodies(asm), resources(rem)
    lines = ["# Hot-kernel ISA facts from a fresh disassembly (tools/hot_kernel_isa.py; hipcc flags of __graft_entry__.py)", ""]
    tot = collections.Counter()
    for b in bodies.values():
        tot.update(counts(b))
    lines.append(f"whole library, {len(bodies)} kernels: scratch/flat/buffer instructions {tot['scratch_flat_buffer']}, v_mfma {tot['mfma']}, "
                 f"SGPR-spill lane operations {tot['sgpr_spill_lane_ops']}, kernels with scratch > 0: {sum(1 for r in res.values() if r.get('scratch'))}, "
                 f"with spilled SGPRs: {sum(1 for r in res.values() if r.get('sspill'))}, with spilled VGPRs: {sum(1 for r in res.values() if r.get('vspill'))}")
    for title, pat in HOT.items():
        name = next(n for n in bodies if pat in n)
        c, r = counts(bodies[name]), res.get(name, {})
        lines += ["", f"## {title}", f"static instruction counts (every path of the kernel, rare branches included): {dict(sorted(c.items()))}",
                  f"registers: {r.get('vgpr')} VGPRs, {r.get('sgpr')} SGPRs, spilled SGPRs {r.get('sspill')}, spilled VGPRs {r.get('vspill')}, "
                  f"scratch {r.get('scratch')} B/lane, LDS {r.get('lds')} B/block, occupancy {r.get('occ')} waves/SIMD",
                  "load block (first vector load .. first wait for vector memory):"]
        lines += ["    " + ln for ln in load_block(bodies[name])]
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    text = report()
    print(text)
    if "--write" in sys.argv:
        tag = sys.argv[sys.argv.index("--write") + 1]
        with open(os.path.join(REPO, "profiles", f"{tag}_hot_kernel_isa.txt"), "w") as f:
            f.write(text)
