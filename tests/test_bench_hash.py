"""bench.py ties the PMC-derived numbers it quotes (roofline.traffic, roofline.valu) to a hash of the kernel sources:
a measurement taken on other CODE must read "stale", but editing a comment must not throw a valid measurement away."""
import json
import os

import bench
from conftest import REPO


def test_hash_ignores_comments_and_white_space_but_not_code(tmp_path, monkeypatch):
    src = 'int f(int a) { // add one\n    /* really */ return a + 1;   }\nconst char* s = "// not a comment /* nor this */";\n'
    same = 'int f(int a) {\n  return a+1; }   // moved comment\nconst char* s = "// not a comment /* nor this */";\n'
    other = src.replace("a + 1", "a + 2")
    string_changed = src.replace("nor this", "nor that")
    assert bench._strip_comments(src) == 'int f(int a) { return a + 1; } const char* s = "// not a comment /* nor this */";'
    files = {}
    for name, text in (("a", src), ("b", same.replace("a+1", "a + 1")), ("c", other), ("d", string_changed)):
        p = tmp_path / f"{name}.h"
        p.write_text(text)
        files[name] = str(p)
    hashes = {}
    for name, path in files.items():
        monkeypatch.setattr(bench, "KERNEL_SOURCES", [path])
        hashes[name] = bench.kernel_source_hash()
    assert hashes["a"] == hashes["b"], "comments / white space must not count"
    assert hashes["a"] != hashes["c"] and hashes["a"] != hashes["d"], "code and string literals must"


def test_committed_pmc_files_carry_a_source_hash():
    for name in ("pmc_traffic.json", "pmc_valu.json"):
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert len(d["kernel_source_sha256_16"]) == 16 and d["source"].startswith("profiles/r0")


def test_a_measurement_is_tied_to_the_built_library(tmp_path, monkeypatch):
    """The counter files belong to the machine code: the hash of fpyv_amd/libfpv_hip.so decides (a compiled-out experiment hook
    changes the sources but not the library); a file from before that hash existed falls back to the source hash."""
    lib = bench.library_hash()
    assert len(lib) == 16 and lib == bench.library_hash()
    assert bench.measurement_is_current({"library_sha256_16": lib, "kernel_source_sha256_16": "0" * 16})
    assert bench.measurement_is_current({"library_sha256_16": "0" * 16, "kernel_source_sha256_16": bench.kernel_source_hash()}), "built elsewhere: the source hash decides"
    assert not bench.measurement_is_current({"library_sha256_16": "0" * 16, "kernel_source_sha256_16": "0" * 16})
    assert bench.measurement_is_current({"kernel_source_sha256_16": bench.kernel_source_hash()})
    assert not bench.measurement_is_current({"kernel_source_sha256_16": "0" * 16})
