"""The Rust shim (rust/fastllm-mi355x) cannot be compiled in this image (no rustc / cargo).  What can be
checked without a Rust compiler is checked here: the #[repr(C)] layouts and the extern "C" surface of src/ffi.rs against
the real C header, and that the patch for the reference tree applies."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "fastllm-mi355x")
HEADER = os.path.join(ROOT, "include", "fastllm_mi355x.h")

# Rust type -> (size, align, C spelling used only for documentation) under the x86-64 / LP64 C ABI
PRIM = {"i32": (4, 4), "u32": (4, 4), "i64": (8, 8), "u64": (8, 8), "f64": (8, 8), "f32": (4, 4), "usize": (8, 8),
        "c_int": (4, 4), "c_char": (1, 1), "u8": (1, 1)}


def rust_structs():
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", src, re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for fm in re.finditer(r"pub (\w+): ([^,\n]+),", body):
            fields.append((fm.group(1), fm.group(2).strip()))
        out[name] = fields
    return out


def layout(fields, structs):
    """C layout of a #[repr(C)] struct: [(name, offset, size)], total size, alignment."""
    off, align, rows = 0, 1, []
    for name, ty in fields:
        arr = re.match(r"\[(\w+); (\d+)\]", ty)
        if ty.startswith("*"):
            size, al = 8, 8
        elif arr:
            es, al = PRIM[arr.group(1)]
            size = es * int(arr.group(2))
        elif ty in PRIM:
            size, al = PRIM[ty]
        else:                                   # nested struct
            _, size, al = layout(structs[ty], structs)
        off = (off + al - 1) // al * al
        rows.append((name, off, size))
        off += size
        align = max(align, al)
    return rows, (off + align - 1) // align * align, align


def test_repr_c_layouts_match_the_c_header():
    structs = rust_structs()
    public = ["fl_config", "fl_tensor", "fl_parallel", "fl_model_info", "fl_sampling", "fl_kernel_stat"]
    for s in public:
        assert s in structs, "src/ffi.rs lacks #[repr(C)] struct %s" % s
    lines = ['#include <stddef.h>', '#include "%s"' % HEADER]
    for s in public:
        rows, size, _ = layout(structs[s], structs)
        lines.append('_Static_assert(sizeof(%s) == %d, "sizeof(%s): Rust mirror says %d");' % (s, size, s, size))
        for name, off, fsz in rows:
            lines.append('_Static_assert(offsetof(%s, %s) == %d, "%s.%s offset");' % (s, name, off, s, name))
            lines.append('_Static_assert(sizeof(((%s *)0)->%s) == %d, "%s.%s size");' % (s, name, fsz, s, name))
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "layout.c")
        open(c, "w").write("\n".join(lines) + "\nint main(void) { return 0; }\n")
        r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", c], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # and no field of the header is missing from the mirror: same field count per struct
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for s in public:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (s, s), hdr, re.S).group(1)
        n_c = len([x for x in body.split(";") if x.strip()])
        assert n_c == len(structs[s]), "%s: %d fields in the header, %d in src/ffi.rs" % (s, n_c, len(structs[s]))


def c_functions():
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(fl_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_extern_block_declares_exactly_the_header_surface():
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    ext = src[src.index('extern "C" {'):]
    rust = {}
    for m in re.finditer(r"pub fn (fl_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", ext, re.S):
        args = [a for a in re.split(r",\s*(?![^()]*\))", m.group(2).strip()) if a.strip()]
        rust[m.group(1)] = len(args)
    c = c_functions()
    assert set(rust) == set(c), "only in Rust: %s; only in C: %s" % (sorted(set(rust) - set(c)), sorted(set(c) - set(rust)))
    for name in c:
        assert rust[name] == c[name], "%s: %d parameters in C, %d in Rust" % (name, c[name], rust[name])


def test_constants_have_the_header_values():
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    c_vals = dict((k, int(v)) for k, v in re.findall(r"\b(FL_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr))
    c_vals.update((k, int(v)) for k, v in re.findall(r"#define\s+(FL_[A-Z0-9_]+)\s+(-?\d+)", hdr))
    r_vals = dict((k, int(v)) for k, v in re.findall(r"pub const (FL_[A-Z0-9_]+): \w+ = (-?\d+);", src))
    assert set(c_vals) <= set(r_vals), sorted(set(c_vals) - set(r_vals))
    for k, v in c_vals.items():
        assert r_vals[k] == v, k


def test_shim_is_complete_no_elisions():
    for rel in ("src/ffi.rs", "src/safe.rs", "src/lib.rs", "build.rs", "reference-tree/src/models/mi355x.rs"):
        text = open(os.path.join(CRATE, rel)).read()
        assert "todo!" not in text and "unimplemented!" not in text and "/* ... */" not in text and "/* … */" not in text, rel
    glue = open(os.path.join(CRATE, "reference-tree/src/models/mi355x.rs")).read()
    for needle in ("impl<const FAMILY: i32> ModelInitializer for Mi355xWithConfig<FAMILY>", "fn initialize_model", "fn initialize_cache",
                   "fn forward", "impl<const FAMILY: i32> ModelArchitecture for Mi355xWithConfig<FAMILY>", "impl ModelCache for Mi355xCache"):
        assert needle in glue, needle


def test_patch_applies_to_the_reference_tree():
    ref = "/root/reference"
    if not os.path.isdir(ref) or not shutil.which("patch"):
        pytest.skip("reference tree (or patch) not present on this machine")
    with tempfile.TemporaryDirectory() as d:
        for f in ("Cargo.toml", "src/main.rs", "src/models/mod.rs", "src/models/model_registry.rs"):
            os.makedirs(os.path.dirname(os.path.join(d, f)), exist_ok=True)
            shutil.copy(os.path.join(ref, f), os.path.join(d, f))
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(CRATE, "patches", "0001-mi355x-backend.patch")],
                           cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
