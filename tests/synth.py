"""Seeded synthetic weights + small model configs shared by the tests, the golden
generator (tests/golden/make_golden.py) and __graft_entry__.smoke().

Weights are N(0, std^2) drawn from numpy's frozen legacy stream
(np.random.RandomState: bit-stable across numpy versions) and ROUNDED TO BF16, so the
same values feed the fp32 oracle, HF transformers (as fp32) and the GPU bf16 path
without any further rounding.  Tensor names are the HF safetensors names the reference
binds by (/root/reference/src/providers/huggingface/huggingface.rs:83-130).
"""
import hashlib
import zlib

import numpy as np


def f32_to_bf16_bits(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


# Small configs.  head_dim is 64 or 128 like the BASELINE models (TinyLlama 64; Mistral /
# Qwen2 128) and GQA group sizes cover 2, 3 and 4.
CONFIGS = {
    "llama_a": dict(family="llama", hidden_size=256, intermediate_size=352, vocab_size=256,
                    num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                    rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512),
    "mistral_a": dict(family="mistral", hidden_size=512, intermediate_size=704, vocab_size=320,
                      num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512,
                      sliding_window=4096),
    # candle masks j + sliding_window < i (window+1 keys, App. A.5); HF keeps i - j < W keys:
    # candle sliding_window = 5  <=>  HF sliding_window = 6
    "mistral_win": dict(family="mistral", hidden_size=512, intermediate_size=704, vocab_size=320,
                        num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                        rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512,
                        sliding_window=5),
    "qwen2_a": dict(family="qwen2", hidden_size=384, intermediate_size=512, vocab_size=300,
                    num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2,
                    rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512,
                    sliding_window=4096, qkv_bias=1),
    # Qwen2 with a window that bites (qwen.rs:49-52 forces use_sliding_window=true): G = 3, q/k/v bias, theta 1e6;
    # candle sliding_window = 5  <=>  HF sliding_window = 6.  Decode steps attend to the whole cache (App. A.5: no mask at T = 1)
    "qwen2_win": dict(family="qwen2", hidden_size=384, intermediate_size=512, vocab_size=300,
                      num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2,
                      rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512,
                      sliding_window=5, qkv_bias=1),
    # MHA (no GQA), num_key_value_heads absent -> defaults to heads (llama.rs:39)
    "llama_mha": dict(family="llama", hidden_size=128, intermediate_size=256, vocab_size=200,
                      num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=None,
                      rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512),
    # head_dim that is not 64 / 128 (the reference takes any even value, config.rs:31-43): 100 (not a multiple of 8; OpenLLaMA-3B's),
    # 96 with GQA 3 + q/k/v bias, 48 (below 64).  The library runs them padded to 128 / 128 / 64 (model.hip, resolve_config).
    "llama_d100": dict(family="llama", hidden_size=400, intermediate_size=352, vocab_size=256,
                       num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512),
    "qwen2_d96": dict(family="qwen2", hidden_size=576, intermediate_size=512, vocab_size=300,
                      num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2,
                      rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512,
                      sliding_window=4096, qkv_bias=1),
    "mistral_d48": dict(family="mistral", hidden_size=192, intermediate_size=256, vocab_size=320,
                        num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                        rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512,
                        sliding_window=4096),
    # enough kv heads for a 4-way tensor-parallel group (no golden fixture: TP tests only)
    "llama_tp4": dict(family="llama", hidden_size=512, intermediate_size=1024, vocab_size=512,
                      num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4,
                      rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512),
    # ... and for an 8-way one (BASELINE config C4's degree): GQA 2, one kv head and 64 vocabulary rows per rank
    "llama_tp8": dict(family="llama", hidden_size=1024, intermediate_size=2048, vocab_size=512,
                      num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=8,
                      rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512),
    # Mistral-7B's layer shape (h 4096, i 14336, 32/8 heads of 128), two layers, small vocabulary: full-size GEMV grids
    "mistral_wide": dict(family="mistral", hidden_size=4096, intermediate_size=14336, vocab_size=1024,
                         num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=8,
                         rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=1024, sliding_window=4096),
}


def tensor_shapes(cfg):
    h, i, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    Hkv = cfg.get("num_key_value_heads") or H
    d = h // H
    bias = bool(cfg.get("qkv_bias", cfg["family"] == "qwen2"))
    out = [("model.embed_tokens.weight", (V, h))]
    for l in range(L):
        p = "model.layers.%d." % l
        out += [(p + "self_attn.q_proj.weight", (H * d, h)),
                (p + "self_attn.k_proj.weight", (Hkv * d, h)),
                (p + "self_attn.v_proj.weight", (Hkv * d, h)),
                (p + "self_attn.o_proj.weight", (h, H * d)),
                (p + "mlp.gate_proj.weight", (i, h)),
                (p + "mlp.up_proj.weight", (i, h)),
                (p + "mlp.down_proj.weight", (h, i)),
                (p + "input_layernorm.weight", (h,)),
                (p + "post_attention_layernorm.weight", (h,))]
        if bias:
            out += [(p + "self_attn.q_proj.bias", (H * d,)),
                    (p + "self_attn.k_proj.bias", (Hkv * d,)),
                    (p + "self_attn.v_proj.bias", (Hkv * d,))]
    out += [("model.norm.weight", (h,)), ("lm_head.weight", (V, h))]
    return out


def synth_weights(cfg, seed=0xFA57, std=0.05, lm_head_scale=4.0):
    """dict name -> uint16 array of bf16 bit patterns."""
    w = {}
    for name, shape in tensor_shapes(cfg):
        rs = np.random.RandomState((seed + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        a = rs.standard_normal(shape).astype(np.float32)
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            a = 1.0 + 0.1 * a
        elif name.endswith(".bias"):
            a = 0.1 * a
        elif name == "lm_head.weight":
            a = std * lm_head_scale * a          # decisive argmax (SURVEY 8d knob)
        else:
            a = std * a
        w[name] = f32_to_bf16_bits(a)
    return w


def as_f32(weights):
    return {k: bf16_bits_to_f32(v) for k, v in weights.items()}


def weights_digest(weights):
    hsh = hashlib.sha256()
    for k in sorted(weights):
        hsh.update(k.encode())
        hsh.update(np.ascontiguousarray(weights[k]).tobytes())
    return hsh.hexdigest()


def prompt_ids(cfg, T, seed=1234):
    rs = np.random.RandomState(seed)
    ids = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    ids[0] = 1                                   # BOS first
    return ids
