"""A/B of the prefill GEMM kernels inside ONE process (same box, same clock history): alternates FL_GEMM_4W=0 / 1 over
repeated prefills of T tokens, back to back (3 per sample) and prints the medians.  usage: prefill_ab.py [model] [T ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
Ts = [int(t) for t in sys.argv[2:]] or [512]
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    res = {"0": [], "1": []}
    for rep in range(7):
        for mode in ("0", "1") if rep % 2 == 0 else ("1", "0"):
            os.environ["FL_GEMM_4W"] = mode
            gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize()
            res[mode].append((time.perf_counter() - t0) / 3 * 1e3)
    m0, m1 = np.median(res["0"]), np.median(res["1"])
    print("%s T=%5d: eight waves %.3f ms (%.3f..%.3f)   four waves %.3f ms (%.3f..%.3f)   ratio %.3f" % (
        name, T, m0, min(res["0"]), max(res["0"]), m1, min(res["1"]), max(res["1"]), m1 / m0), flush=True)
    c.close()
