"""Per-kernel profile (HIP events per launch) of decode steps at full model shapes.  usage: decode_profile.py [model] [dtype] [prompt] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 512
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cfg = dict(MODEL_CONFIGS[name])
os.environ["FL_GRAPH"] = "0"                      # per-launch events need plain launches
fa.reload_env()
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype=dtype)
del wts; torch.cuda.empty_cache()
p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
c = gm.new_cache(T + steps + 8)
tok = gm.forward_argmax(c, p, 0)
gm.forward_argmax(c, np.array([tok], dtype=np.uint32), T)
gm.profile_begin()
for i in range(steps):
    tok = gm.forward_argmax(c, np.array([tok], dtype=np.uint32), T + 1 + i)
st = gm.profile_end()
tot = sum(s["total_ms"] for s in st)
print("%s %s decode at %d: %.3f ms per step in kernels" % (name, dtype, T, tot / steps))
for s in sorted(st, key=lambda s: -s["total_ms"]):
    print("  %-44s x%-4d %8.3f ms/step  %7.1f us/launch  %s" % (s["name"], s["launches"] // steps, s["total_ms"] / steps, s["total_ms"] * 1e3 / s["launches"],
          ("%.2f TB/s" % (s["bytes"] / s["total_ms"] / 1e9)) if s.get("bytes") else ""))
