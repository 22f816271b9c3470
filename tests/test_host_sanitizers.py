"""The host layer's input parsers under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).

The checkpoint reader (fastllm_amd/host/safetensors.hpp) stands where candle's `safetensors` crate stands in the reference
(/root/reference/src/providers/huggingface/huggingface.rs:83-130): it takes bytes from disk and hands pointers + shapes to
fl_model_create, so a header it mis-validates becomes a read past the mapping on the way to HBM.  tests/host/fuzz_host_inputs.cc
is built with -fsanitize=address,undefined -fno-sanitize-recover and fed hand-made hostile files plus hypothesis-generated
mutations of a valid checkpoint; every input must be accepted or rejected with an Error, and nothing may trip a sanitizer.
"""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import synth
from test_host_mirror import config_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "fuzz_host_inputs")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined,float-cast-overflow,float-divide-by-zero", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "host", "fuzz_host_inputs.cc"), "-o", exe])
    return exe


def run(driver, paths):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([driver] + [str(p) for p in paths], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "fuzz driver done" in out.stdout, "sanitizer report or crash:\n" + out.stdout[-2000:] + out.stderr[-6000:]
    verdicts = {}
    for ln in out.stdout.splitlines():
        f = ln.split(" ", 3)
        if f[0] in ("accepted", "rejected"):
            verdicts[(f[1], f[2])] = (f[0], f[3] if len(f) > 3 else "")
    return verdicts


def st_file(path, header, data=b"", header_len=None):
    """a safetensors file from a header dict (or raw header bytes) and a data section"""
    hb = header if isinstance(header, bytes) else json.dumps(header).encode()
    n = len(hb) if header_len is None else header_len
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", n) + hb + data)
    return path


def entry(dtype, shape, b, e):
    return {"dtype": dtype, "shape": shape, "data_offsets": [b, e]}


def test_hostile_safetensors_headers_are_rejected(driver, tmp_path):
    d = tmp_path
    good = st_file(d / "good.safetensors", {"a": entry("F32", [2, 3], 0, 24), "b": entry("BF16", [4], 24, 32), "__metadata__": {"format": "pt"}}, bytes(range(32)))
    empty_tensor = st_file(d / "empty_tensor.safetensors", {"a": entry("F32", [0, 7], 0, 0), "b": entry("F16", [2], 0, 4)}, b"\x01\x02\x03\x04")
    bad = {
        # the round-4 finding: cnt *= (size_t)d.num wraps -- 2^32 x 2^32 x 1 elements "fit" in 4 bytes
        "dims_overflow": st_file(d / "dims_overflow.safetensors", {"a": entry("F32", [4294967296, 4294967296, 1], 0, 4)}, b"\0" * 4),
        "dims_overflow2": st_file(d / "dims_overflow2.safetensors", {"a": entry("BF16", [9223372036854775807, 2], 0, 4)}, b"\0" * 4),
        "dims_huge_double": st_file(d / "dims_huge_double.safetensors", b'{"a":{"dtype":"F32","shape":[1e300],"data_offsets":[0,4]}}', b"\0" * 4),
        "dims_negative": st_file(d / "dims_negative.safetensors", {"a": entry("F32", [-1, -4], 0, 16)}, b"\0" * 16),
        "dims_fraction": st_file(d / "dims_fraction.safetensors", b'{"a":{"dtype":"F32","shape":[1.5,2],"data_offsets":[0,12]}}', b"\0" * 12),
        "dims_nan": st_file(d / "dims_nan.safetensors", b'{"a":{"dtype":"F32","shape":[nan],"data_offsets":[0,4]}}', b"\0" * 4),
        "dims_string": st_file(d / "dims_string.safetensors", b'{"a":{"dtype":"F32","shape":["4"],"data_offsets":[0,16]}}', b"\0" * 16),
        "too_many_dims": st_file(d / "too_many_dims.safetensors", {"a": entry("F32", [1] * 40, 0, 4)}, b"\0" * 4),
        "offsets_negative": st_file(d / "offsets_negative.safetensors", {"a": entry("F32", [1], -4, 0)}, b"\0" * 4),
        "offsets_reversed": st_file(d / "offsets_reversed.safetensors", {"a": entry("F32", [1], 4, 0)}, b"\0" * 4),
        "offsets_past_end": st_file(d / "offsets_past_end.safetensors", {"a": entry("F32", [4], 0, 16)}, b"\0" * 8),
        "offsets_huge": st_file(d / "offsets_huge.safetensors", b'{"a":{"dtype":"F32","shape":[1],"data_offsets":[18446744073709551612,18446744073709551616]}}', b"\0" * 4),
        "offsets_overlap": st_file(d / "offsets_overlap.safetensors", {"a": entry("F32", [2], 0, 8), "b": entry("F32", [2], 4, 12)}, b"\0" * 12),
        "offsets_hole": st_file(d / "offsets_hole.safetensors", {"a": entry("F32", [1], 0, 4), "b": entry("F32", [1], 8, 12)}, b"\0" * 12),
        "trailing_bytes": st_file(d / "trailing_bytes.safetensors", {"a": entry("F32", [1], 0, 4)}, b"\0" * 12),
        "shape_mismatch": st_file(d / "shape_mismatch.safetensors", {"a": entry("F32", [3], 0, 8)}, b"\0" * 8),
        "bad_dtype": st_file(d / "bad_dtype.safetensors", {"a": entry("I64", [1], 0, 8)}, b"\0" * 8),
        "duplicate": st_file(d / "duplicate.safetensors", b'{"a":{"dtype":"F32","shape":[1],"data_offsets":[0,4]},"a":{"dtype":"F32","shape":[1],"data_offsets":[4,8]}}', b"\0" * 8),
        "header_len_huge": st_file(d / "header_len_huge.safetensors", b"{}", header_len=1 << 40),
        "header_len_max": st_file(d / "header_len_max.safetensors", b"{}", header_len=(1 << 64) - 1),
        "header_truncated": st_file(d / "header_truncated.safetensors", b'{"a":{"dtype":"F32","shape":[1],"data_off'),
        "header_not_object": st_file(d / "header_not_object.safetensors", b"[1,2,3]"),
        "header_deep": st_file(d / "header_deep.safetensors", b"[" * 200000),
        "header_deep_obj": st_file(d / "header_deep_obj.safetensors", b'{"a":' * 100000),
        "header_bad_escape": st_file(d / "header_bad_escape.safetensors", b'{"a\\u12'),
        "entry_not_object": st_file(d / "entry_not_object.safetensors", b'{"a":[1,2]}'),
        "entry_missing_offsets": st_file(d / "entry_missing_offsets.safetensors", b'{"a":{"dtype":"F32","shape":[1]}}', b"\0" * 4),
    }
    for name, size in (("short7", 7), ("empty", 0)):
        p = d / (name + ".safetensors")
        p.write_bytes(b"\x01" * size)
        bad[name] = p
    v = run(driver, [good, empty_tensor] + list(bad.values()))
    assert v[("safetensors", str(good))][0] == "accepted" and "2 tensors, byte sum %d" % sum(range(32)) in v[("safetensors", str(good))][1]
    assert v[("safetensors", str(empty_tensor))][0] == "accepted"
    for name, p in bad.items():
        assert v[("safetensors", str(p))][0] == "rejected", (name, v[("safetensors", str(p))])


def test_hostile_index_and_config_files(driver, tmp_path):
    from test_safetensors import write_checkpoint
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    ok = tmp_path / "ok"
    write_checkpoint(str(ok), cfg, w, shards=3)
    cases = {"ok": (ok, "accepted")}

    def ckdir(name, index):
        p = tmp_path / name
        p.mkdir()
        (p / "model.safetensors.index.json").write_bytes(index if isinstance(index, bytes) else json.dumps(index).encode())
        return p
    cases["escape"] = (ckdir("escape", {"weight_map": {"a": "../ok/model-00001-of-00003.safetensors"}}), "rejected")
    cases["absolute"] = (ckdir("absolute", {"weight_map": {"a": str(ok / "model-00001-of-00003.safetensors")}}), "rejected")
    cases["missing_shard"] = (ckdir("missing_shard", {"weight_map": {"a": "nope.safetensors"}}), "rejected")
    cases["no_weight_map"] = (ckdir("no_weight_map", {"metadata": {}}), "rejected")
    cases["index_truncated"] = (ckdir("index_truncated", b'{"weight_map": {"a": "x.safe'), "rejected")
    cases["index_deep"] = (ckdir("index_deep", b'{"weight_map":' + b"[" * 100000), "rejected")
    cases["nothing"] = (ckdir("nothing", {"weight_map": {}}), "accepted")            # no shards named: an empty map (initialize_model then misses its tensors)
    (tmp_path / "void").mkdir()
    cases["void"] = (tmp_path / "void", "rejected")
    v = run(driver, [p for p, _ in cases.values()])
    for name, (p, want) in cases.items():
        assert v[("checkpoint", str(p))][0] == want, (name, v[("checkpoint", str(p))])

    good_cfg = tmp_path / "config_good.json"
    good_cfg.write_bytes(config_json(cfg))
    cfgs = {"good": (good_cfg, "accepted")}
    for name, patch in (("neg_hidden", {"hidden_size": -64}), ("frac_layers", {"num_hidden_layers": 1.5}), ("huge_vocab", {"vocab_size": 1e300}),
                        ("neg_kv", {"num_key_value_heads": -2}), ("zero_heads", {"num_attention_heads": 0}), ("string_hidden", {"hidden_size": "256"}),
                        ("neg_window", {"sliding_window": -1}), ("odd_head_dim", {"hidden_size": 12, "num_attention_heads": 4})):
        j = json.loads(config_json(cfg))
        j.update(patch)
        p = tmp_path / ("config_%s.json" % name)
        p.write_text(json.dumps(j))
        cfgs[name] = (p, "rejected")
    for name, raw in (("cfg_truncated", b'{"hidden_size": 25'), ("cfg_array", b"[]"), ("cfg_empty", b""), ("cfg_unterminated", b'{"a": "xx'),
                      ("cfg_nul", b'{"hidden_size":\x00 1}'), ("cfg_deep", b'{"a":' + b"[" * 50000)):
        p = tmp_path / (name + ".json")
        p.write_bytes(raw)
        cfgs[name] = (p, "rejected")
    v = run(driver, [p for p, _ in cfgs.values()])
    for name, (p, want) in cfgs.items():
        assert v[("config", str(p))][0] == want, (name, v[("config", str(p))])


def test_mutated_checkpoints_never_trip_the_sanitizers(driver, tmp_path):
    """hypothesis: byte-level and field-level mutations of a valid file.  The driver reads every byte of every tensor it accepts,
    so an accepted header that lies about its shape is an ASan report, not a pass."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    rs = np.random.RandomState(5)
    data = rs.bytes(64)
    base = {"w": entry("F32", [2, 4], 0, 32), "b": entry("BF16", [16], 32, 64), "__metadata__": {"format": "pt"}}
    batch, expect_ok = [], []

    ints = st.one_of(st.integers(-2 ** 70, 2 ** 70), st.sampled_from([0, 1, 2 ** 31, 2 ** 32, 2 ** 53, 2 ** 63, 2 ** 64 - 1]),
                     st.floats(allow_nan=True, allow_infinity=True))

    @settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck), database=None)
    @given(shape=st.lists(ints, max_size=5), off=st.tuples(ints, ints), cut=st.integers(0, 200), flip=st.integers(0, 400), mode=st.integers(0, 3))
    def gen(shape, off, cut, flip, mode):
        i = len(batch)
        p = tmp_path / ("m%04d.safetensors" % i)
        hdr = json.loads(json.dumps(base))
        if mode == 0:                                   # lying fields
            hdr["w"]["shape"], hdr["w"]["data_offsets"] = shape, list(off)
            blob = struct.pack("<Q", 0)
            hb = json.dumps(hdr, allow_nan=True).encode()
            blob = struct.pack("<Q", len(hb)) + hb + data
        else:
            hb = json.dumps(hdr).encode()
            blob = bytearray(struct.pack("<Q", len(hb)) + hb + data)
            if mode == 1:
                blob = blob[:max(0, len(blob) - cut)]   # truncation
            elif mode == 2:
                blob[flip % len(blob)] ^= 1 << (flip % 8)   # one flipped bit (header length, JSON or data)
            else:
                blob[8 + flip % len(hb)] = ord("[{\"}]:,-e0"[flip % 10])
        p.write_bytes(bytes(blob))
        batch.append(p)
    gen()
    assert len(batch) >= 100
    v = run(driver, batch)
    assert len(v) == len(batch)
    # sanity of the corpus: it is not all garbage -- flipped data bits leave valid files
    assert any(x[0] == "accepted" for x in v.values()) and any(x[0] == "rejected" for x in v.values())
