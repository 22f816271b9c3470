"""A/B of a decode-path switch inside ONE process (same box, same clock history): alternates two values of an environment
variable; every sample is a fresh cache (so a fresh hipGraph capture under that value), the same prompt, K greedy steps timed.
usage: AB_ENV=FL_ATTN_PREFETCH AB_A=0 AB_B=1 python tools/decode_ab.py [model] [prompt] [steps]
(AB_TUNE=gemv_u instead of AB_ENV: the two values go through fl_tune)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ENV, VA, VB = os.environ.get("AB_ENV", "FL_ATTN_PREFETCH"), os.environ.get("AB_A", "0"), os.environ.get("AB_B", "1")
TUNE = os.environ.get("AB_TUNE")
if TUNE: ENV = "fl_tune(%s)" % TUNE
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
prompt = np.random.RandomState(1234).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
prompt[0] = 1
res, toks = {VA: [], VB: []}, {}
for rep in range(6):
    for mode in (VA, VB) if rep % 2 == 0 else (VB, VA):
        if TUNE: fa.tune(TUNE, int(mode))
        else: os.environ[ENV] = mode
        c = gm.new_cache(T + K + 96)
        first = gm.forward_argmax(c, prompt, 0)
        t = gm.decode_greedy(c, first, T, 16)                  # warm-up + capture
        gm.synchronize(); t0 = time.perf_counter()
        t2 = gm.decode_greedy(c, int(t[-1]), T + 16, K)
        gm.synchronize()
        res[mode].append(K / (time.perf_counter() - t0))
        toks[mode] = np.concatenate([t, t2])
        c.close()
ma, mb = np.median(res[VA]), np.median(res[VB])
print("%s prompt %d, %d steps: %s=%s %.1f tokens/s (%.1f..%.1f)   %s=%s %.1f tokens/s (%.1f..%.1f)   ratio %.4f   ids equal: %s" % (
    name, T, K, ENV, VA, ma, min(res[VA]), max(res[VA]), ENV, VB, mb, min(res[VB]), max(res[VB]), mb / ma, bool(np.array_equal(toks[VA], toks[VB]))))
