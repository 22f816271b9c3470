#!/usr/bin/env python3
"""One FULL-DEPTH run of the CPU baseline (oracle/ref_forward.c, all 32 layers of the Mistral-7B shape, the bench's
synthetic weights): validates bench.py's 2-of-32 / 8-of-32 layer extrapolation, and times it at the workload's own kv
length (512-token prompt) next to the 16-token one.  Needs ~30 GB of host memory (bf16 weights kept as bf16)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from fastllm_amd.configs import MODEL_CONFIGS
from oracle import oracle
cfg = MODEL_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
ext = bench.cpu_baseline(torch, cfg, wts)
print("extrapolated (bench.py):", ext["value"], "tokens/s;", ext["sample"])
host = {k: v.view(torch.int16).cpu().numpy().view(np.uint16) for k, v in wts.items()}
del wts
om = oracle.OracleModel(cfg, host, threads=oracle.default_threads())
for kv in (16, 512):
    oc = om.new_cache(kv + 16)
    prompt = np.arange(1, kv + 1, dtype=np.uint32) % cfg["vocab_size"]
    t0 = time.perf_counter()
    tok = oracle.argmax(om.forward(oc, prompt, 0))
    tp = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(8):
        tok = oracle.argmax(om.forward(oc, [tok], kv + i))
    dt = (time.perf_counter() - t0) / 8
    print("full depth, %d layers, %d threads, kv_len %d..%d: %.3f tokens/s decode (%.1f ms/token); %d-token prefill %.1f s"
          % (cfg["num_hidden_layers"], oracle.default_threads(), kv, kv + 8, 1.0 / dt, dt * 1e3, kv, tp), flush=True)
