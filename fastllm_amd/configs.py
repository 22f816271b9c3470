"""config.json fields of the BASELINE.json models.  TinyLlama's dims are
also literally in the reference (src/models/llama.rs:126-141)."""

MODEL_CONFIGS = {
    "tinyllama-1.1b": dict(family="llama", hidden_size=2048, intermediate_size=5632, vocab_size=32000,
                           num_hidden_layers=22, num_attention_heads=32, num_key_value_heads=4,
                           rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=2048),
    "mistral-7b": dict(family="mistral", hidden_size=4096, intermediate_size=14336, vocab_size=32000,
                       num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                       rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768,
                       sliding_window=4096),
    "qwen2-7b": dict(family="qwen2", hidden_size=3584, intermediate_size=18944, vocab_size=152064,
                     num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                     rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=32768,
                     sliding_window=131072, qkv_bias=1),
}


def decode_bytes_per_token(cfg, kv_len, bytes_per_elem=2):
    """Algorithmic HBM bytes of one decoded token:
    weights read once + K and V of kv_len cached positions."""
    h, i, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    Hkv = cfg.get("num_key_value_heads") or H
    d = h // H
    bias = (h + 2 * Hkv * d) if cfg.get("qkv_bias") else 0
    w = L * (2 * h * h + 2 * Hkv * d * h + 3 * h * i + 2 * h + bias) + h + V * h
    return bytes_per_elem * (w + L * Hkv * d * 2 * kv_len)


def prefill_flops(cfg, T):
    """Algorithmic FLOPs of a T-token prefill."""
    h, i, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    Hkv = cfg.get("num_key_value_heads") or H
    d = h // H
    return 2 * T * L * (2 * h * h + 2 * Hkv * d * h + 3 * h * i) + 2 * V * h + L * 2 * T * T * h
