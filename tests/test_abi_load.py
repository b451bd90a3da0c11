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
    return sorted(set(re.findall(r"\b(fpv_[a-z_0-9]+)\s*\(", hdr)))


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
    assert L.fpv_state_rows(0) == 14 and L.fpv_state_rows(1) == 29 and L.fpv_state_rows(7) == -1
    assert L.fpv_state_rows(1) == _lib.FPV_RACER_ROWS
    assert L.fpv_algorithmic_bytes(0) == 133 and L.fpv_algorithmic_bytes(1) == 181     # SURVEY 8d figures
    assert _lib.algorithmic_bytes(0) == 133 and _lib.algorithmic_bytes(1) == 181


def test_checkpoint_labels_are_the_librarys_own():
    """The strings a checkpoint records for its fp16 storage words and its stick-noise stream are defined next to the code they
    describe (csrc/fpv_math.h) and exported (fpv_encoding_id): the Python copies cannot drift from the kernels'."""
    from fpyv_amd import env
    L = _lib.lib()
    assert L.fpv_encoding_id(0).decode() == env.STATE_H_ENCODING and L.fpv_encoding_id(1).decode() == env.NOISE_GENERATOR
    assert L.fpv_encoding_id(2) is None


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


def test_lane_offsets_fit_32_bits_up_to_the_drone_limit():
    """The widest lane-addressed element is the 16-byte action row: 16 * i must not wrap below the
    per-handle limit, fpv_create must refuse anything above it (this check needs no device), and the
    arithmetic at i = 2^28 + 1 shows why."""
    from oracle import lane_model
    Lm = lane_model.lib()
    Lm.fpvl_lane_offset.restype = C.c_uint32
    Lm.fpvl_lane_offset.argtypes = [C.c_uint32, C.c_uint32]
    Lm.fpvl_max_drones.restype = C.c_int64
    limit = Lm.fpvl_max_drones()
    assert limit == 1 << 28
    for elem in (1, 2, 4, 8, 16):
        for i in (0, 1, 12345, limit - 1):
            assert Lm.fpvl_lane_offset(i, elem) == i * elem                     # exact below the limit
    assert Lm.fpvl_lane_offset(limit + 1, 16) != (limit + 1) * 16              # would wrap: hence the limit
    assert Lm.fpvl_lane_offset(limit + 1, 16) == 16
    L = _lib.lib()
    p = _lib.pack_params(load_params(fps=1000))
    h = C.c_void_p()
    rc = L.fpv_create(C.byref(p), limit + 1, 0, C.byref(h))
    assert rc == -1 and b"2^28" in L.fpv_last_error() and not h.value


def _l2_set_overflow(stride_bytes, blocks, rows=14, ways=16):
    """the model of fpv_hip.hip `l2_set_overflow`, restated: of the lines of `blocks` drone blocks x 14 rows that ONE XCD touches
    (every eighth block: 512 B of every 4 KiB of a row), the fraction beyond the 16 ways of its set, set = (L ^ (L >> 11)) & 2047"""
    import numpy as np
    j = np.arange(blocks // 8, dtype=np.int64)[None, :, None]
    r = np.arange(rows, dtype=np.int64)[:, None, None]
    line = np.arange(4, dtype=np.int64)[None, None, :]
    L = (r * stride_bytes + j * 4096 + line * 128) >> 7
    worst = 0.0
    for sets in (L ^ (L >> 11), L + (L >> 11), L - (L >> 11)):          # the fold and its additive twins (the measured penalty is symmetric)
        cnt = np.bincount((sets & 2047).ravel(), minlength=2048)
        worst = max(worst, float(np.maximum(cnt - ways, 0).sum()) / L.size)
    return worst


def test_recommended_row_stride_rules():
    """fpv_recommended_ld is host arithmetic (no device): any n gets a stride >= n of whole 16-byte groups that keeps 1 KiB clear of
    a multiple of 8 KiB; beyond 2^18 drones it is 1 KiB past a multiple of 2 KiB (the best class of stride at every measured size,
    profiles/r05_exp_row_stride_l2_sets.log) unless that makes the rows of a block meet in the same L2 sets - 2^19 drones with the
    former pad of 256 floats: 71 % of the lines an XCD should keep do not fit their set (13.2 against 10.7 us per launch)."""
    L = _lib.lib()
    for n in (1, 63, 64, 300, 4096, 70001, 1 << 17, 1 << 18):                       # the small-population rule is the one of rounds 1-4
        old = (n + 63) // 64 * 64
        old += (256 - old % 2048) if old % 2048 < 256 else 0
        assert L.fpv_recommended_ld(n) == old
    import random
    rng = random.Random(5)
    sizes = [1 << 19, 3 << 18, 1 << 20, 3 << 19, 1 << 21, 5 << 19, 1 << 23, 1 << 28, 1_000_000, 750_000, 2_000_000, 540_672] + [rng.randrange(1 << 18, 1 << 24) for _ in range(40)]
    for n in sizes:
        ld = L.fpv_recommended_ld(n)
        assert n <= ld < n + 1024 and ld % 4 == 0 and ld % 2048 >= 256 and ld % 512 != 0, (n, ld)
        if n > (1 << 21):
            assert ld % 512 == 256, (n, ld)
        elif n > (1 << 18):
            blocks = min((n + 1023) // 1024 * 8, 4096)
            preferred = (n + 255) // 512 * 512 + 256
            if _l2_set_overflow(4 * preferred, blocks) < 0.06:
                assert ld == preferred, (n, ld)
            else:
                assert _l2_set_overflow(4 * ld, blocks) < _l2_set_overflow(4 * preferred, blocks), (n, ld)
    assert L.fpv_recommended_ld(1 << 20) == (1 << 20) + 256 and L.fpv_recommended_ld(1 << 21) == (1 << 21) + 256
    assert L.fpv_recommended_ld(1 << 19) == (1 << 19) + 320 and L.fpv_recommended_ld(3 << 19) == (3 << 19) + 320
    assert _l2_set_overflow(4 * ((1 << 19) + 256), 4096) > 0.7 and _l2_set_overflow(4 * ((1 << 19) + 320), 4096) < 0.01
    assert _l2_set_overflow(4 * ((1 << 20) + 256), 4096) < 0.01 and _l2_set_overflow(4 * (1 << 20), 4096) > 0.4
    assert L.fpv_recommended_ld(0) < 0 and L.fpv_recommended_ld(-5) < 0


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


def test_components_module_mirrors_reference_names_and_constructor_signatures():
    """fpyv_amd.components takes the reference's own constructor calls (components.py:646, :686-694,
    :754-763, :73; simulator.py:53-58): positional rendering arguments are accepted and ignored."""
    import numpy as np
    import fpyv_amd.components as c
    from fpyv_amd.env import DroneBatch, RacerBatch
    from fpyv_amd.objects import to_rows
    assert c.Drone is DroneBatch and c.Racer is RacerBatch
    g = c.Ground(size=60, resolution=50, random=True)                                   # params.yaml:6-9
    assert g.as_row()[0] == 0 and c.Ground(60, 4).as_row() == g.as_row() and c.Ground().as_row() == g.as_row()
    cyl = c.Cylinder(np.array([1, 2, 0]), 0.5, 2.0, 10, 25, random=True)                # generators.py:33-37
    assert cyl.as_row() == (1, 1.0, 2.0, 0.0, 0.5, 2.0) and c.Cylinder([1, 2, 0], 0.5, 2.0).as_row() == cyl.as_row()
    t = c.Target(np.array([0, 0, 3.0]), 0.8, 5, {"radius": 2.0, "resolution": 8})       # generators.py:22-25
    t.update()
    assert t.as_row()[:4] == (2, 2.0, 0.0, 3.0)          # first path point: centre + (radius, 0, 0)
    assert c.Target([0, 0, 3], 0.8, path={"radius": 2.0, "resolution": 8}).radius == 0.8
    with pytest.raises(AssertionError):
        c.Cylinder([0, 0, 0], -1.0, 2.0, 4, 2)            # components.py:688
    # Gate and Trail entries of an object_list never collide in the reference (components.py:202): skipped
    gate = c.Gate(np.zeros(3), np.eye(3), 5.0, shape="circle", resolution=17)
    trail = c.Trail(10)
    rows = to_rows([t, gate, cyl, trail, g])
    assert [r[0] for r in rows] == [2, 1, 0]
    assert c.PID is not None


def test_vec_env_refuses_an_unknown_mode():
    """Checked before anything touches the GPU or the library (the reference raises ValueError on its mode strings too,
    components.py:278,301)."""
    from fpyv_amd.env import FpvVecEnv
    with pytest.raises(ValueError, match="mode must be"):
        FpvVecEnv(num_envs=4, mode="plane")
