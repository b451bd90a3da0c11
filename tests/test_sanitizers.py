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
