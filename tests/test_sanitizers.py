"""AddressSanitizer + UBSan over the two CPU builds (the float64 oracle and the host build of the
kernel's lane arithmetic): `make -C oracle asan` compiles both with -fsanitize=address,undefined and runs
oracle/sanitize_driver.cpp, which calls every entry point on ragged batches in every variant."""
import os
import subprocess

from conftest import REPO


def test_cpu_builds_are_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizers: clean" in r.stdout


def test_product_library_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    """The PRODUCT library's host code that needs no device - the row-stride rule with its L2 set model, argument validation, error
    text - under AddressSanitizer + UBSan: libfpv_hip.so is rebuilt with every
    `-fsanitize=` behind `-Xarch_host` (device code as always) and tests/host_sanitize/driver.c runs against it.  (With a GPU the
    launch logic runs under UBSan through the whole GPU suite: tools/gpu/host_ubsan.sh; ROCm's ASan runtime cannot live in a process
    that initialises the device.)"""
    import sys
    sys.path.insert(0, REPO)
    from __graft_entry__ import HIPCC_FLAGS
    hipcc, clang = "/opt/rocm/bin/hipcc", "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.isfile(hipcc) and os.path.isfile(clang)):
        import pytest
        pytest.skip("no ROCm toolchain")
    lib = os.path.join(tmp_path, "libfpv_hip.so")
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize-recover=undefined",
           "-Xarch_host", "-fno-omit-frame-pointer", "-g"]
    flags = [f for f in HIPCC_FLAGS if f != "-O3"] + ["-O1"]
    r = subprocess.run([hipcc] + flags + san + ["-o", lib, os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(tmp_path, "driver")
    r = subprocess.run([clang, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", exe,
                        os.path.join(REPO, "tests", "host_sanitize", "driver.c"), "-L" + str(tmp_path), "-lfpv_hip", "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "host sanitizers: clean" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
