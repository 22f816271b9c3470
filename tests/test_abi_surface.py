"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, fails loudly without a GPU (no CPU fallback), and the pure-host entry points work."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fastllm_mi355x.h")


@pytest.fixture(scope="module")
def fa():
    import fastllm_amd
    return fastllm_amd


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fl_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(fa):
    names = declared_functions()
    assert len(names) >= 20, names
    L = fa.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/fastllm_mi355x.h but not exported: %s" % missing
    out = subprocess.check_output(["nm", "-D", "--defined-only", fa.binding.LIB_PATH], text=True)
    exported = set(re.findall(r" T (fl_[a-z_0-9]+)", out))
    assert set(names) <= exported
    assert fa.abi_version() == 2


def test_library_contains_gfx950_code_objects(fa):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", fa.binding.LIB_PATH],
                         capture_output=True, text=True)
    blob = out.stdout + out.stderr
    if "gfx" not in blob:      # older objdump: fall back to the bundle marker in the binary
        blob = open(fa.binding.LIB_PATH, "rb").read().decode("latin1")
    assert "gfx950" in blob


def test_struct_layouts_match_header(fa):
    b = fa.binding
    assert C.sizeof(b.FlConfig) == 8 + 8 * 8 + 16
    assert C.sizeof(b.FlTensor) == 8 + 8 + 32 + 8 + 8
    assert C.sizeof(b.FlParallel) == 16 + 8 + 8
    assert C.sizeof(b.FlKernelStat) == 48 + 8 + 8 + 8 + 8


def test_no_cpu_fallback(fa):
    """Without a GPU the product must fail loudly, not compute on the host."""
    if fa.device_count() > 0:
        pytest.skip("a GPU is visible here")
    cfg = synth.CONFIGS["llama_a"]
    with pytest.raises(fa.FastLLMError) as e:
        fa.Model(cfg, synth.synth_weights(cfg))
    assert e.value.code == -9 and "no CPU path" in str(e.value)          # FL_ERR_NO_DEVICE
    x = np.zeros((1, 64), np.float32)
    with pytest.raises(fa.FastLLMError) as e:
        fa.op_linear(x, np.zeros((8, 64), np.float32))
    assert e.value.code == -9


def test_config_validation_errors_precede_device_probe(fa):
    bad = dict(synth.CONFIGS["llama_a"], num_attention_heads=3)         # 256 % 3 != 0
    with pytest.raises(fa.FastLLMError) as e:
        fa.Model(bad, {})
    assert e.value.code == -1 and "divisible" in str(e.value)
    bad = dict(synth.CONFIGS["llama_a"], num_key_value_heads=3)
    with pytest.raises(fa.FastLLMError) as e:
        fa.Model(bad, {})
    assert e.value.code == -1


def test_tp_slice_partition(fa):
    """fl_tp_slice: column-parallel q/k/v/gate/up/lm_head, row-parallel o/down; the shards tile the
    tensor exactly once."""
    cfg = synth.CONFIGS["mistral_a"]          # h 512, H 4, Hkv 2, d 128, i 704, V 320
    for tp in (1, 2):
        for name, shape in synth.tensor_shapes(cfg):
            cover = np.zeros(shape if len(shape) == 2 else (shape[0], 1), dtype=np.int32)
            for r in range(tp):
                r0, r1, c0, c1 = fa.tp_slice(cfg, name, r, tp)
                sl = (slice(r0, r1), slice(c0, c1))
                cover[sl] += 1
            replicated = ("layernorm" in name) or name in ("model.norm.weight", "model.embed_tokens.weight")
            assert (cover == (tp if replicated else 1)).all(), name
    assert fa.tp_slice(cfg, "model.layers.0.self_attn.q_proj.weight", 1, 2) == (256, 512, 0, 512)
    assert fa.tp_slice(cfg, "model.layers.0.self_attn.k_proj.weight", 1, 2) == (128, 256, 0, 512)
    assert fa.tp_slice(cfg, "model.layers.0.self_attn.o_proj.weight", 1, 2) == (0, 512, 256, 512)
    assert fa.tp_slice(cfg, "model.layers.0.mlp.down_proj.weight", 0, 2) == (0, 512, 0, 352)
    assert fa.tp_slice(cfg, "lm_head.weight", 1, 2) == (160, 320, 0, 512)
    with pytest.raises(fa.FastLLMError):
        fa.tp_slice(cfg, "model.layers.0.self_attn.q_proj.weight", 0, 3)      # 3 does not divide the heads
    with pytest.raises(fa.FastLLMError):
        fa.tp_slice(cfg, "not.a.tensor", 0, 1)


def test_algorithmic_bytes_formula(fa):
    """decode weight bytes per token for the BASELINE models."""
    from fastllm_amd.configs import MODEL_CONFIGS, decode_bytes_per_token, prefill_flops
    assert abs(decode_bytes_per_token(MODEL_CONFIGS["tinyllama-1.1b"], 0) / 1e9 - 2.0690) < 2e-3
    assert abs(decode_bytes_per_token(MODEL_CONFIGS["mistral-7b"], 0) / 1e9 - 14.2213) < 2e-3
    assert abs(decode_bytes_per_token(MODEL_CONFIGS["qwen2-7b"], 0) / 1e9 - 14.1412) < 2e-3
    assert decode_bytes_per_token(MODEL_CONFIGS["mistral-7b"], 1) - decode_bytes_per_token(MODEL_CONFIGS["mistral-7b"], 0) == 131072
    assert abs(prefill_flops(MODEL_CONFIGS["mistral-7b"], 512) / 1e12 - 7.216) < 0.01
    assert abs(prefill_flops(MODEL_CONFIGS["qwen2-7b"], 4096) / 1e12 - 56.82) < 0.05


def test_exception_barrier_at_the_abi(fa, monkeypatch):
    """Nothing throws across the C ABI (the host is Rust: an unwinding C++ exception there is UB; the reference
    surfaces failures as anyhow::Error, mod.rs:402-405).  FL_DEBUG_THROW makes model_create throw where a failed
    `new` / container growth would: bad_alloc -> FL_ERR_OOM, anything else -> FL_ERR_HIP, message set, process alive."""
    cfg = synth.CONFIGS["llama_a"]
    for kind, code, text in (("bad_alloc", -4, "bad_alloc"), ("runtime", -5, "injected failure"), ("int", -5, "unknown C++ exception")):
        monkeypatch.setenv("FL_DEBUG_THROW", "model_create=" + kind)
        with pytest.raises(fa.FastLLMError) as e:
            fa.Model(cfg, {})
        assert e.value.code == code and text in str(e.value), (kind, str(e.value))
    monkeypatch.setenv("FL_DEBUG_THROW", "cache_create=bad_alloc")     # another site's name: model_create is unaffected
    with pytest.raises(fa.FastLLMError) as e:
        fa.Model(cfg, {})
    assert e.value.code in (-9, -2)                                     # no device here / no tensors on a GPU box
    monkeypatch.delenv("FL_DEBUG_THROW")


def test_null_out_pointers_are_rejected(fa):
    L = fa.lib()
    b = fa.binding
    c = b.FlConfig()
    assert L.fl_model_create(C.byref(c), None, 0, 1, None, None) == -8      # FL_ERR_BAD_ARGUMENT, not a segfault
    assert b"null out" in L.fl_last_error()
    assert L.fl_cache_create(None, 16, None) == -8
    assert L.fl_device_count(None) == -8
