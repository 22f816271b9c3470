"""Which single allocation, filled with a finite garbage byte, changes the logits?  usage: poison_scan.py model dtype [byte] [max index]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
import fastllm_amd as fa
name, dtype = sys.argv[1], sys.argv[2]
byte = int(sys.argv[3]) if len(sys.argv) > 3 else 63
nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 70
cfg = synth.CONFIGS[name]
w = synth.synth_weights(cfg)
ids = synth.prompt_ids(cfg, 14, seed=11)
def run():
    g = fa.Model(cfg, w, dtype=dtype)
    c = g.new_cache(64)
    out = [g.forward(c, ids[:10], 0)] + [g.forward(c, ids[i:i + 1], i) for i in range(10, 12)]
    c.close(); g.close()
    return out
fa.tune("debug_poison", 0)
ref = run()
for n in range(0, nmax):
    fa.tune("debug_poison", byte | (n << 8))
    got = run()
    d = [float(np.abs(a - b).max()) if np.isfinite(a).all() else float("nan") for a, b in zip(got, ref)]
    if any(x != 0 for x in d):
        print("allocation %d: prefill %.3g decode %.3g %.3g" % (n, d[0], d[1], d[2]), flush=True)
fa.tune("debug_poison", 0)
print("scan done")
