"""Row N1: the safetensors reader on the input side of initialize_model
(reference: providers/huggingface/huggingface.rs:83-130), host-only tests."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import synth
from test_host_mirror import config_json, host  # noqa: F401


def write_checkpoint(dirname, cfg, weights_bits, shards=1):
    """HF-style directory: config.json + model.safetensors (or shards + index), bf16 tensors."""
    import torch
    from safetensors.torch import save_file
    os.makedirs(dirname, exist_ok=True)
    open(os.path.join(dirname, "config.json"), "wb").write(config_json(cfg))
    tens = {k: torch.from_numpy(v.view(np.int16).copy()).view(torch.bfloat16) for k, v in weights_bits.items()}
    if shards == 1:
        save_file(tens, os.path.join(dirname, "model.safetensors"), metadata={"format": "pt"})
    else:
        names = sorted(tens)
        wm = {}
        for s in range(shards):
            fn = "model-%05d-of-%05d.safetensors" % (s + 1, shards)
            part = {k: tens[k] for k in names[s::shards]}
            save_file(part, os.path.join(dirname, fn), metadata={"format": "pt"})
            wm.update({k: fn for k in part})
        json.dump({"metadata": {"total_size": 0}, "weight_map": wm}, open(os.path.join(dirname, "model.safetensors.index.json"), "w"))


def probe(host, dirname, name):
    host.flh_checkpoint_probe.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    n, dt, nd, s = C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_uint64(0)
    shape = (C.c_int64 * 4)()
    rc = host.flh_checkpoint_probe(dirname.encode(), name.encode() if name else None, C.byref(n), C.byref(dt), C.byref(nd), shape, C.byref(s))
    return rc, n.value, dt.value, tuple(shape[: nd.value]), s.value


@pytest.mark.parametrize("shards", [1, 3])
def test_reader_matches_independent_reader(host, tmp_path, shards):
    from safetensors import safe_open
    cfg = synth.CONFIGS["qwen2_a"]
    w = synth.synth_weights(cfg)
    d = str(tmp_path / ("ckpt%d" % shards))
    write_checkpoint(d, cfg, w, shards)
    rc, n, _, _, _ = probe(host, d, None)
    assert rc == 0 and n == len(w)
    for name in ("model.embed_tokens.weight", "model.layers.1.self_attn.k_proj.bias", "model.layers.0.mlp.down_proj.weight", "lm_head.weight"):
        rc, n, dt, shape, bsum = probe(host, d, name)
        assert rc == 0, host.flh_last_error()
        assert dt == 1 and shape == w[name].shape                    # BF16
        assert bsum == int(w[name].view(np.uint8).astype(np.uint64).sum())
    # cross-check one tensor's bytes with the safetensors package itself
    files = [f for f in os.listdir(d) if f.endswith(".safetensors")]
    found = False
    for f in files:
        with safe_open(os.path.join(d, f), framework="pt") as sf:
            if "model.norm.weight" in sf.keys():
                import torch
                ref = sf.get_tensor("model.norm.weight").view(torch.int16).numpy().view(np.uint16)
                np.testing.assert_array_equal(ref, w["model.norm.weight"])
                found = True
    assert found


def test_reader_errors(host, tmp_path):
    d = str(tmp_path / "empty")
    os.makedirs(d)
    rc, *_ = probe(host, d, None)
    assert rc != 0 and b"model.safetensors" in host.flh_last_error()         # huggingface.rs:95
    bad = str(tmp_path / "bad")
    os.makedirs(bad)
    open(os.path.join(bad, "model.safetensors"), "wb").write((1 << 40).to_bytes(8, "little") + b"{}")
    rc, *_ = probe(host, bad, None)
    assert rc != 0 and b"header length" in host.flh_last_error()
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    ok = str(tmp_path / "ok")
    write_checkpoint(ok, cfg, w)
    rc, *_ = probe(host, ok, "model.layers.9.mlp.up_proj.weight")
    assert rc == -2                                                          # FL_ERR_MISSING_TENSOR


def test_load_dir_rejects_cpu_and_unknown_arch(host, tmp_path):
    host.flh_load_dir.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    cfg = synth.CONFIGS["llama_a"]
    d = str(tmp_path / "ck")
    write_checkpoint(d, cfg, synth.synth_weights(cfg))
    h, fam = C.c_void_p(), C.c_int(-1)
    assert host.flh_load_dir(d.encode(), 1, -1, C.byref(h), C.byref(fam)) == -9       # no CPU path
    j = json.loads(config_json(cfg))
    j["architectures"] = ["GPT2LMHeadModel"]
    json.dump(j, open(os.path.join(d, "config.json"), "w"))
    assert host.flh_load_dir(d.encode(), 1, 0, C.byref(h), C.byref(fam)) == -1 and b"Unsupported architecture" in host.flh_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("name,shards", [("llama_a", 1), ("mistral_a", 2), ("qwen2_a", 1)])
def test_load_dir_and_generate(host, tmp_path, name, shards, monkeypatch):
    """config.json + safetensors on disk -> registry family pick -> initialize_model -> generate == oracle."""
    from oracle import oracle
    from test_gpu_host_mirror import generate
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    host.flh_load_dir.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    d = str(tmp_path / name)
    write_checkpoint(d, cfg, w, shards)
    h, fam = C.c_void_p(), C.c_int(-1)
    assert host.flh_load_dir(d.encode(), 0, 0, C.byref(h), C.byref(fam)) == 0, host.flh_last_error()   # fp32 compute
    assert fam.value == {"llama": 0, "mistral": 1, "qwen2": 2}[cfg["family"]]
    om = oracle.OracleModel(cfg, synth.as_f32(w))
    prompt = synth.prompt_ids(cfg, 8)
    want = om.generate(om.new_cache(64), prompt, 10, pos_mode="reference")
    got, _ = generate(host, h, prompt, 10)
    np.testing.assert_array_equal(got, want)
    host.flh_model_destroy(h)
