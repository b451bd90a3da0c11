"""The C-ABI shared library loads without a GPU and exports every symbol include/fpv_abi.h
declares; argument validation that needs no device works; compute entry points fail loudly
(FPV_ENODEV) instead of falling back to the CPU."""
import ctypes as C
import os
import re

import pytest

from conftest import REPO
from fpyv_amd import _lib, load_params


def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "fpv_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fpv_[a-z_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), f"libfpv_hip.so does not export {name}"
    assert L.fpv_abi_version() == _lib.FPV_ABI_VERSION


def test_struct_sizes_match_the_c_side():
    # struct_size is checked by fpv_create; a mismatch must be reported, not ignored
    L = _lib.lib()
    p = _lib.pack_params(load_params(fps=1000))
    p.struct_size += 8
    h = C.c_void_p()
    rc = L.fpv_create(C.byref(p), 16, 0, C.byref(h))
    assert rc < 0 and not h.value
    assert L.fpv_state_rows(0) == 14 and L.fpv_state_rows(1) == 20 and L.fpv_state_rows(7) == -1
    assert L.fpv_algorithmic_bytes(0) == 133 and L.fpv_algorithmic_bytes(1) == 181
    assert _lib.algorithmic_bytes(0) == 133 and _lib.algorithmic_bytes(1) == 181


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    p = _lib.pack_params(load_params(fps=1000))
    h = C.c_void_p()
    rc = L.fpv_create(C.byref(p), 1024, 0, C.byref(h))
    assert rc == -3 and L.fpv_error_name(rc) == b"FPV_ENODEV"
    assert b"HIP device" in L.fpv_last_error()
    with pytest.raises((ValueError, _lib.FpvError, RuntimeError)):
        from fpyv_amd.env import DroneBatch
        DroneBatch(load_params(fps=1000), 8, device="cuda:0")
    with pytest.raises(ValueError, match="GPU only"):
        from fpyv_amd.env import DroneBatch
        DroneBatch(load_params(fps=1000), 8, device="cpu")


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under fpyv_amd/ or include/ may reference it."""
    bad = []
    for root in ("fpyv_amd", "include"):
        for d, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle|fpvo_|fpvl_", txt, flags=re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_components_module_mirrors_reference_names():
    import fpyv_amd.components as c
    from fpyv_amd.env import DroneBatch, RacerBatch
    assert c.Drone is DroneBatch and c.Racer is RacerBatch
    assert c.Ground().as_row()[0] == 0 and c.Cylinder([1, 2, 0], 0.5, 2.0).as_row() == (1, 1.0, 2.0, 0.0, 0.5, 2.0)
    t = c.Target([0, 0, 3], 0.8, path={"radius": 2.0, "resolution": 8})
    t.update()
    assert t.as_row()[:4] == (2, 2.0, 0.0, 3.0)          # first path point: centre + (radius, 0, 0)
    with pytest.raises(AssertionError):
        c.Cylinder([0, 0, 0], -1.0, 2.0)                  # components.py:688
