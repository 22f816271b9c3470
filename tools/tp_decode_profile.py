"""Per-kernel HIP-event profile of decode steps of a tensor-parallel model in the EMULATED mode (every rank's shard on this one
GPU, one after the other): the per-rank kernel shapes and times of a tp-way group, without its links.
usage: tp_decode_profile.py [model] [tp] [prompt]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
kw = {} if tp == 1 else dict(tp_mode=fa.binding.TP_EMULATED, tp_size=tp)
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", **kw)
del wts; torch.cuda.empty_cache()
p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
c = gm.new_cache(T + 64)
first = gm.forward_argmax(c, p, 0)
toks = gm.decode_greedy(c, first, T, 8)
gm.profile_begin()
gm.decode_greedy(c, int(toks[-1]), T + 8, 8)
st = gm.profile_end()
tot = sum(s["total_ms"] for s in st)
print("%s tp=%d S=%d: %.3f ms per step in kernels (all %d shards; %.3f per rank)" % (name, tp, T, tot / 8, tp, tot / 8 / tp))
for s in sorted(st, key=lambda s: -s["total_ms"]):
    print("  %-40s x%-5d %8.2f us/launch  %7.1f GB/s" % (s["name"], s["launches"] // 8, s["total_ms"] * 1e3 / s["launches"],
          s["bytes"] / (s["total_ms"] * 1e-3) / 1e9 if s["total_ms"] else 0))
