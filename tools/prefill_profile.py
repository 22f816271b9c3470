"""Per-kernel profile (HIP events per launch) of one prefill at full model shapes.  usage: prefill_profile.py [model] [T] [layers] [tp]
(tp > 1: FL_TP_EMULATED -- every rank's shard on this one GPU, one after the other: the per-rank kernel shapes of a tensor-parallel group)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = dict(MODEL_CONFIGS[name])
if len(sys.argv) > 3:
    cfg["num_hidden_layers"] = int(sys.argv[3])
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
tp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
DT = os.environ.get("PP_DTYPE", "bf16")
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype=DT) if tp == 1 else fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", tp_mode=fa.binding.TP_EMULATED, tp_size=tp)
del wts; torch.cuda.empty_cache()
p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
c = gm.new_cache(T + 8)
gm.forward_argmax(c, p, 0); c.reset()
gm.profile_begin(); gm.forward_argmax(c, p, 0); st = gm.profile_end()
tot = sum(s["total_ms"] for s in st)
print("%s T=%d layers=%d: %.3f ms in kernels" % (name, T, cfg["num_hidden_layers"], tot))
for s in sorted(st, key=lambda s: -s["total_ms"]):
    print("  %-44s x%-4d %8.3f ms  %7.1f us/launch  %s" % (s["name"], s["launches"], s["total_ms"], s["total_ms"] * 1e3 / s["launches"],
          ("%.0f TFLOP/s" % (s["flops"] / s["total_ms"] / 1e9)) if s["flops"] else ""))
