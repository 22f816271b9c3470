"""What continuous batching costs per membership change: fl_batch_replace against destroying and rebuilding the batch (Mistral-7B, B streams).
usage: replace_cost.py [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = MODEL_CONFIGS["mistral-7b"]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
T = 128
caches, firsts = [], []
for i in range(B + 1):
    c = gm.new_cache(T + 400)
    firsts.append(gm.forward_argmax(c, rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32), 0))
    caches.append(c)
spare, fs = caches.pop(), firsts.pop()
bt = fa.Batch(gm, caches)
pos = [T] * B
g = bt.decode(firsts, pos, 8); toks = [int(x[-1]) for x in g]; pos = [p + 8 for p in pos]
gm.synchronize(); t0 = time.perf_counter()
g = bt.decode(toks, pos, 32); gm.synchronize()
step = (time.perf_counter() - t0) / 32
toks = [int(x[-1]) for x in g]; pos = [p + 32 for p in pos]
# (a) replace slot 3 back and forth, one step after each swap
cur, other, pcur, pother, tcur, tother = caches[3], spare, pos[3], T, toks[3], fs
t0 = time.perf_counter()
N = 20
for k in range(N):
    bt.replace(3, other)
    cur, other, pcur, pother, tcur, tother = other, cur, pother, pcur, tother, tcur
    toks[3], pos[3] = tcur, pcur
    g = bt.decode(toks, pos, 1)
    toks = [int(x[-1]) for x in g]; pos = [p + 1 for p in pos]
    tcur, pcur = toks[3], pos[3]
gm.synchronize()
a = (time.perf_counter() - t0) / N
# (b) the same membership change by rebuilding the batch
t0 = time.perf_counter()
for k in range(N):
    bt.close()
    lst = list(caches); lst[3] = other
    cur, other, pcur, pother, tcur, tother = other, cur, pother, pcur, tother, tcur
    caches = lst
    toks[3], pos[3] = tcur, pcur
    bt = fa.Batch(gm, caches)
    g = bt.decode(toks, pos, 1)
    toks = [int(x[-1]) for x in g]; pos = [p + 1 for p in pos]
    tcur, pcur = toks[3], pos[3]
gm.synchronize()
b = (time.perf_counter() - t0) / N
print("mistral-7b, %d streams: a decode step %.3f ms; membership change + one step: fl_batch_replace %.3f ms, rebuild (destroy + create) %.3f ms" % (B, step * 1e3, a * 1e3, b * 1e3))
