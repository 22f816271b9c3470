"""The oracle is test infrastructure: nothing in the product may import, link or call it."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product_files():
    for base in ("fastllm_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cc", ".c", "Makefile")):
                    yield os.path.join(d, f)


def test_product_never_references_the_oracle():
    pat = re.compile(r"oracle|orc_|ref_forward|liboracle")
    bad = []
    for p in _product_files():
        for i, line in enumerate(open(p, errors="replace")):
            if pat.search(line) and "no CPU" not in line:
                bad.append("%s:%d: %s" % (os.path.relpath(p, ROOT), i + 1, line.strip()))
    assert not bad, "\n".join(bad)


def test_product_library_does_not_link_the_oracle():
    so = os.path.join(ROOT, "fastllm_amd", "lib", "libfastllm_mi355x.so")
    out = subprocess.check_output(["ldd", so], text=True)
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", so], text=True)
    assert "orc_" not in syms


def test_no_reference_sources_in_tree():
    # the reference is Rust: no .rs text may live in the repo except the illustrative shim in INTEGRATION.md
    for d, _, files in os.walk(ROOT):
        if ".git" in d or "gpurun_out" in d:
            continue
        for f in files:
            assert not f.endswith(".rs"), os.path.join(d, f)


def test_required_layout_exists():
    for p in ("bench.py", "__graft_entry__.py", "include/fastllm_mi355x.h", "oracle/ref_forward.c",
              "tests/golden/make_golden.py", "fastllm_amd/csrc/k_gemv.hip", "fastllm_amd/host/fastllm_host.hpp"):
        assert os.path.exists(os.path.join(ROOT, p)), p
