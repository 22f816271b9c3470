"""Prefill time over prompt lengths that are NOT the BASELINE's round numbers (tile / block quantisation shows here).
Usage (GPU box): python tools/prefill_ragged.py mistral-7b [T ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
Ts = [int(t) for t in sys.argv[2:]] or [1100, 1536, 2000, 2048, 2500, 3000, 3072, 3500, 4000, 4096, 4100, 4608, 5000, 6000, 6144, 7000, 8192]
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
h, i, L, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["vocab_size"]
d = h // cfg["num_attention_heads"]; hkv = cfg.get("num_key_value_heads", cfg["num_attention_heads"])
for T in Ts:
    p = rs.randint(0, V, size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    gm.forward_argmax(c, p, 0); c.reset()
    gm.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        gm.forward_argmax(c, p, 0); c.reset()
    gm.synchronize(); dt = (time.perf_counter() - t0) / 2
    flops = 2.0 * T * L * (2 * h * h + 2 * hkv * d * h + 3 * h * i) + 2.0 * V * h + L * 2.0 * T * T * h
    print("T=%5d: %8.2f ms  %8.0f tokens/s  %5.1f %% of 2.5 PFLOP/s" % (T, dt * 1e3, T / dt, flops / dt / 2.5e15 * 100), flush=True)
    c.close()
