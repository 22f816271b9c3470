"""Prefill latency vs prompt length (Mistral-7B shape): where does the T>1 path sit relative to one weight read?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    gm.forward_argmax(c, p, 0); c.reset()
    gm.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        gm.forward_argmax(c, p, 0); c.reset()
    gm.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("T=%5d: %8.3f ms  %9.1f tokens/s" % (T, dt * 1e3, T / dt), flush=True)
    c.close()
