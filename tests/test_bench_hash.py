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
