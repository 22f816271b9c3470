"""Short prompts and decode batches with / without the opt-in five-launch layer (FL_GEMM_SKF 2 / 1), one process, alternating, medians.
usage: short_ab.py model T[,T...] [batch sizes B,B]
FL_GEMM_SKF=2 acts in the EXPERIMENTAL build only: run with FL_LIB_PATH=fastllm_amd/lib/libfastllm_mi355x_exp.so
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
Ts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,16,33,64,100,128").split(",")]
Bs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    res = {0: [], 1: []}
    for rep in range(7):
        for mode in (1, 0):
            fa.tune("gemm_skf", 2 if mode else 1)
            c.reset(); gm.forward_argmax(c, p, 0); c.reset()
            gm.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                c.reset(); gm.forward_argmax(c, p, 0)
            gm.synchronize()
            res[mode].append((time.perf_counter() - t0) / 3)
    a, b = sorted(res[1])[3] * 1e3, sorted(res[0])[3] * 1e3
    print("%s prefill T=%4d: five-launch layer %.3f ms   default %.3f ms   x%.3f" % (name, T, a, b, a / b), flush=True)
    c.close()
for B in Bs:
    T, K = 512, 48
    out = {}
    for mode in (1, 0):
        fa.tune("gemm_skf", 2 if mode else 1)
        caches, firsts = [], []
        for i in range(B):
            ci = gm.new_cache(T + 2 * K + 80)
            firsts.append(gm.forward_argmax(ci, rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32), 0))
            caches.append(ci)
        bt = fa.Batch(gm, caches)
        g = bt.decode(firsts, [T] * B, 8)
        gm.synchronize(); t0 = time.perf_counter()
        g = bt.decode([int(x[-1]) for x in g], [T + 8] * B, K)
        gm.synchronize()
        out[mode] = (time.perf_counter() - t0) / K * 1e3
        bt.close()
        for ci in caches:
            ci.close()
    print("%s batch B=%2d: five-launch layer %.3f ms/step (%.0f tok/s)   default %.3f ms/step (%.0f tok/s)   x%.3f" % (name, B, out[1], B / out[1] * 1e3, out[0], B / out[0] * 1e3, out[1] / out[0]), flush=True)
fa.tune("reload_env", 0)
