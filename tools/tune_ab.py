"""A/B of a fl_tune switch inside ONE process (same box, same clock history): alternates two values over repeated prefills of T tokens,
back to back (3 per sample), and prints the medians.  usage: tune_ab.py model key A B [T ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name, key, VA, VB = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
Ts = [int(t) for t in sys.argv[5:]] or [512]
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    res, tok = {VA: [], VB: []}, {}
    for rep in range(7):
        for mode in (VA, VB) if rep % 2 == 0 else (VB, VA):
            fa.tune(key, mode)
            tok[mode] = gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize()
            res[mode].append((time.perf_counter() - t0) / 3 * 1e3)
    m0, m1 = np.median(res[VA]), np.median(res[VB])
    print("%s T=%5d: %s=%d %.3f ms (%.3f..%.3f)   %s=%d %.3f ms (%.3f..%.3f)   ratio %.3f   first token %s" % (
        name, T, key, VA, m0, min(res[VA]), max(res[VA]), key, VB, m1, min(res[VB]), max(res[VB]), m1 / m0,
        "same" if tok[VA] == tok[VB] else "DIFFERS (%d / %d)" % (tok[VA], tok[VB])), flush=True)
    c.close()
