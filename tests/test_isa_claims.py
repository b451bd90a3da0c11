"""The kernel-quality claims of DESIGN.md against a FRESH gfx950 disassembly (hipcc cross-compiles here, no GPU needed):
no scratch, no MFMA, no spilled register anywhere in the library, and the instruction counts / register numbers DESIGN
quotes for the hot kernels.  tools/hot_kernel_isa.py is the same code the committed profiles/r06_hot_kernel_isa.txt was
written with."""
import os
import re
import shutil
import sys

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    import hot_kernel_isa as h
    asm, rem = h.disassemble(str(tmp_path_factory.mktemp("isa") / "fpv.s"))
    return h, h.kernel_bodies(asm), h.resources(rem)


def _design():
    return open(os.path.join(REPO, "DESIGN.md"), encoding="utf-8").read()


def test_no_scratch_no_mfma_no_spill_in_any_kernel(isa):
    h, bodies, res = isa
    assert len(bodies) == 42 and len(res) >= 42
    for name, body in bodies.items():
        c = h.counts(body)
        assert c["scratch_flat_buffer"] == 0, name
        assert c["mfma"] == 0, name               # there is no dense contraction on this path
        assert c["sgpr_spill_lane_ops"] == 0, name
    for name, r in res.items():
        assert r.get("scratch", 0) == 0 and r.get("sspill", 0) == 0 and r.get("vspill", 0) == 0, (name, r)
    assert "42 kernels" in _design()


def test_plain_step_kernel_counts_as_design_quotes(isa):
    """DESIGN 2 / 3.1 and csrc/fpv_addr.h: 14 state rows through v_lshl_add_u64 address pairs + vaddr loads, the action
    through the saddr form, 24 stores, three v_med3 clips, no LDS; 74 VGPRs / 6 waves."""
    h, bodies, res = isa
    name = next(n for n in bodies if h.HOT["plain single-step kernel fpv_drone_step_kernel<false,false,false,false>"] in n)
    body, c, r = bodies[name], h.counts(bodies[name]), res[name]
    assert c["global_load"] == 21 and c["global_store"] == 24 and c["v_med3_f32"] == 3 and c["lds"] == 0
    block = h.load_block(body)
    state_loads = [ln for ln in block if re.match(r"global_load_dword v\d+, v\[\d+:\d+\], off", ln)]
    assert len(state_loads) == 14, "the 14 fp32 state rows: 64-bit address in a VGPR pair"
    assert sum(1 for ln in block if ln.startswith("v_lshl_add_u64")) == 14
    assert any(re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\] nt", ln) for ln in block), "action row: saddr form, non-temporal"
    assert sum(1 for ln in block if re.match(r"global_load_dword v\d+, v\d+, s\[\d+:\d+\] nt", ln)) == 4, "SoA sticks: four saddr loads"
    assert (r["vgpr"], r["occ"]) == (73, 6)
    d = _design()
    assert "73 VGPRs (6 waves per SIMD)" in d and "14 `v_lshl_add_u64`" in d


def test_fp16_and_kstep_kernels_as_design_quotes(isa):
    h, bodies, res = isa
    hk = next(n for n in bodies if "fpv_drone_step_h_kernel" in n)
    kk = next(n for n in bodies if h.HOT["plain k-step kernel fpv_drone_rollout_kernel<false,false,false,true>"] in n)
    assert res[hk]["occ"] == 8 and res[hk]["sspill"] == 0
    assert (res[kk]["vgpr"], res[kk]["occ"], res[kk]["sspill"]) == (56, 8, 0)
    nk = next(n for n in bodies if "fpv_drone_rollout_kernelILb1ELb0ELb0ELb1E" in n)
    cn = h.counts(bodies[nk])
    assert cn["lds"] >= 4 and res[nk]["lds"] == 2048, "the noise kernels read the inverse-CDF table from LDS (one ds_read_b128 per normal)"
    assert any(ln.startswith("ds_read_b128") for ln in bodies[nk])


def test_committed_isa_profile_matches_the_sources(isa):
    """profiles/r06_hot_kernel_isa.txt was written by the same tool: its count lines must still be what the sources give."""
    h, bodies, res = isa
    text = open(os.path.join(REPO, "profiles", "r06_hot_kernel_isa.txt")).read()
    for title, pat in h.HOT.items():
        name = next(n for n in bodies if pat in n)
        assert str(dict(sorted(h.counts(bodies[name]).items()))) in text, f"{title}: profiles/r06_hot_kernel_isa.txt is stale (python tools/hot_kernel_isa.py --write r06)"
