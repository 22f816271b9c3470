"""Whole prefills with / without the gate/up row split (FL_GATEUP_ROWSPLIT 1 / 0), one process, alternating, medians.
usage: rowsplit_ab.py [model] [T,T,...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
Ts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "512,513,545,600,700,768,769,1025,1100,1280,1281").split(",")]
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    res = {0: [], 1: []}
    for rep in range(5):
        for mode in (1, 0):
            fa.tune("gateup_rowsplit", mode)
            c.reset(); gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                c.reset(); gm.forward_argmax(c, p, 0)
            gm.synchronize()
            res[mode].append((time.perf_counter() - t0) / 2)
    a, b = sorted(res[1])[2] * 1e3, sorted(res[0])[2] * 1e3
    print("%s prefill T=%4d: row split %.3f ms   one launch %.3f ms   x%.3f" % (name, T, a, b, a / b), flush=True)
    c.close()
fa.tune("reload_env", 0)
