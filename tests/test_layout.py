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
    # the reference is Rust: the only .rs text in the repo is this build's own shim crate (rust/fastllm-mi355x: FFI mirror,
    # safe handles and the NEW mi355x.rs for the reference tree), none of it a copy of a reference file
    ours = os.path.join(ROOT, "rust", "fastllm-mi355x")
    for d, _, files in os.walk(ROOT):
        if ".git" in d or "gpurun_out" in d or d.startswith(ours):
            continue
        for f in files:
            assert not f.endswith(".rs"), os.path.join(d, f)
    ref = "/root/reference/src"
    if os.path.isdir(ref):
        names = set()
        for d, _, files in os.walk(ref):
            names.update(f for f in files if f.endswith(".rs"))
        for d, _, files in os.walk(ours):
            for f in files:
                assert f not in names, "%s shares its name with a reference source file" % os.path.join(d, f)


def test_required_layout_exists():
    for p in ("bench.py", "__graft_entry__.py", "include/fastllm_mi355x.h", "oracle/ref_forward.c",
              "tests/golden/make_golden.py", "fastllm_amd/csrc/k_gemv.hip", "fastllm_amd/host/fastllm_host.hpp"):
        assert os.path.exists(os.path.join(ROOT, p)), p
