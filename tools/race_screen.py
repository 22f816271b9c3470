"""Repeat a long prefill and demand bit-identical logits every time (all kernels are deterministic): a rare wrong
tile from a misplaced wait / barrier shows up as a mismatch or a NaN."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import fastllm_amd as fa
import synth
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = MODEL_CONFIGS[name]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=12)
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
p = synth.prompt_ids(cfg, T, seed=6)
first = None
bad = 0
chunks = [int(x) for x in os.environ.get("SCREEN_CHUNKS", "8192").split(",")]
for r in range(reps):
    if os.environ.get("SCREEN_POLLUTE", "1") == "1":          # freed HBM full of NaN patterns: an uninitialised read shows
        junk = torch.full((6 << 30,), float("nan"), device="cuda", dtype=torch.bfloat16)
        torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    os.environ["FL_PREFILL_CHUNK"] = str(chunks[r % len(chunks)])
    fa.reload_env()
    c = gm.new_cache(T + 16)
    a = gm.forward(c, p, 0)
    c.close()
    if not np.isfinite(a).all():
        print("rep %d: non-finite logits (%d NaN)" % (r, int(np.isnan(a).sum())), flush=True); bad += 1; continue
    if first is None:
        first = a
    elif len(chunks) == 1 and not np.array_equal(a, first):
        print("rep %d: differs from rep 0: max |diff| %g" % (r, np.abs(a - first).max()), flush=True); bad += 1
print("%s T=%d: %d / %d repetitions bad" % (name, T, bad, reps))
