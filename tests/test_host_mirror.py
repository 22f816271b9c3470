"""The C++ host mirror of the reference's trait layer (fastllm_amd/host/fastllm_host.hpp)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fastllm_amd", "lib")


def test_cpp_unit_tests_of_the_mirror(tmp_path):
    """Restatements of the reference's own unit tests (cache.rs, config.rs, llama.rs, mistral.rs, qwen.rs,
    model_registry.rs) against the mirrored C++ types."""
    exe = str(tmp_path / "test_host_mirror")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "host", "test_host_mirror.cc"),
                           "-o", exe, "-L" + LIBDIR, "-lfastllm_mi355x", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror tests ok" in out.stdout


@pytest.fixture(scope="module")
def host():
    import fastllm_amd  # noqa: F401  (loads libfastllm_mi355x.so first)
    L = C.CDLL(os.path.join(LIBDIR, "libfastllm_host.so"))
    L.flh_last_error.restype = C.c_char_p
    L.flh_get_family.restype = C.c_char_p
    L.flh_get_family.argtypes = [C.c_int]
    L.flh_supports_architecture.argtypes = [C.c_int, C.c_char_p]
    L.flh_config_check.argtypes = [C.c_int, C.c_char_p, C.c_void_p]
    L.flh_model_create.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.flh_model_destroy.argtypes = [C.c_void_p]
    L.flh_model_destroy.restype = None
    L.flh_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, C.c_void_p,
                               C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.flh_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.flh_cache_offset.argtypes = [C.c_void_p]
    L.flh_cache_offset.restype = C.c_size_t
    L.flh_cache_reset.argtypes = [C.c_void_p]
    L.flh_cache_reset.restype = None
    return L


def config_json(cfg):
    d = {k: v for k, v in cfg.items() if k not in ("family", "qkv_bias") and v is not None}
    d["architectures"] = [{"llama": "LlamaForCausalLM", "mistral": "MistralForCausalLM", "qwen2": "Qwen2ForCausalLM"}[cfg["family"]]]
    d["torch_dtype"] = "bfloat16"
    return json.dumps(d).encode()


def test_family_registry_strings(host):
    assert [host.flh_get_family(i) for i in range(3)] == [b"Llama", b"Mistral", b"Qwen"]
    assert host.flh_supports_architecture(0, b"LlamaForCausalLM") == 1
    assert host.flh_supports_architecture(0, b"Qwen2ForCausalLM") == 0
    assert host.flh_supports_architecture(2, b"Qwen2_5_VLForConditionalGeneration") == 1


def test_config_json_and_validation(host):
    from fastllm_amd import binding
    out = binding.FlConfig()
    assert host.flh_config_check(1, config_json(synth.CONFIGS["mistral_a"]), C.byref(out)) == 0
    assert (out.hidden_size, out.num_key_value_heads, out.sliding_window, out.family) == (512, 2, 4096, 1)
    bad = dict(synth.CONFIGS["mistral_a"], num_key_value_heads=3)
    assert host.flh_config_check(1, config_json(bad), None) == -101          # assert! -> panic (mistral.rs:109-112)
    assert b"divisible" in host.flh_last_error()
    assert host.flh_config_check(2, config_json(dict(synth.CONFIGS["qwen2_a"], hidden_size=390)), None) == -101   # expect() (qwen.rs:32-37)
    assert host.flh_config_check(0, b'{"hidden_size": 64}', None) == -1      # serde: missing field


def test_initialize_model_rejects_cpu_device(host):
    h = C.c_void_p()
    rc = host.flh_model_create(0, config_json(synth.CONFIGS["llama_a"]), None, 0, 1, -1, C.byref(h))
    assert rc == -9 and b"no CPU path" in host.flh_last_error()
