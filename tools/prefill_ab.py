"""A/B of a per-call switch inside ONE process (same box, same clock history): alternates two values of an environment variable
(default FL_GEMM_4W=0 / 1: the eight-wave / four-wave prefill GEMM) over repeated prefills of T tokens, back to back (3 per sample),
and prints the medians.  usage: prefill_ab.py [model] [T ...]   (AB_ENV=NAME AB_A=value AB_B=value pick another switch)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
Ts = [int(t) for t in sys.argv[2:]] or [512]
ENV, VA, VB = os.environ.get("AB_ENV", "FL_GEMM_4W"), os.environ.get("AB_A", "0"), os.environ.get("AB_B", "1")
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    res = {VA: [], VB: []}
    for rep in range(7):
        for mode in (VA, VB) if rep % 2 == 0 else (VB, VA):
            os.environ[ENV] = mode
            fa.reload_env()
            gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize()
            res[mode].append((time.perf_counter() - t0) / 3 * 1e3)
    m0, m1 = np.median(res[VA]), np.median(res[VB])
    print("%s T=%5d: %s=%s %.3f ms (%.3f..%.3f)   %s=%s %.3f ms (%.3f..%.3f)   ratio %.3f" % (
        name, T, ENV, VA, m0, min(res[VA]), max(res[VA]), ENV, VB, m1, min(res[VB]), max(res[VB]), m1 / m0), flush=True)
    c.close()
