import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, synth
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1]
cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=int(sys.argv[2]))
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=13)
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts
for T in [int(t) for t in sys.argv[3:]]:
    ids = synth.prompt_ids(cfg, T, seed=19)
    res = {}
    for mode, resid in ((0, 0), (1, 0), (1, 0), (1, 1)):
        fa.tune("gemm_h4", mode); fa.tune("gemm_resid", resid)
        c = gm.new_cache(T + 8)
        lg = gm.forward(c, ids, 0)
        lg2 = gm.forward(c, ids[:1], T)
        c.close()
        res.setdefault((mode, resid), []).append((lg, lg2))
    ref = res[(0, 0)][0]
    for k, v in res.items():
        for n, (a, b) in enumerate(v):
            print("%s T=%d h4=%d resid=%d run %d: prefill rel %.2e  decode rel %.2e" % (name, T, k[0], k[1], n,
                  np.linalg.norm(a - ref[0]) / np.linalg.norm(ref[0]), np.linalg.norm(b - ref[1]) / np.linalg.norm(ref[1])), flush=True)
