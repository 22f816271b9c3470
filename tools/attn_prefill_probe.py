"""Prefill attention kernel time (HIP events around each launch) at full model shapes, few layers.
usage: attn_prefill_probe.py [model] [T ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b"
Ts = [int(x) for x in sys.argv[2:]] or [512, 2048, 4096]
cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=4)
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
H, d = cfg["num_attention_heads"], cfg["hidden_size"] // cfg["num_attention_heads"]
for T in Ts:
    p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    gm.forward_argmax(c, p, 0); c.reset()
    gm.profile_begin(); tok = gm.forward_argmax(c, p, 0); st = gm.profile_end()
    a = [s for s in st if s["name"].startswith("attn_prefill")][0]
    per = a["total_ms"] / a["launches"]
    fl = 2.0 * T * T * H * d                       # causal half of QK^T and PV
    print("%s T=%5d: attn_prefill %.3f ms/layer  %.1f TFLOP/s (causal)  first token %d" % (name, T, per, fl / per / 1e9, tok), flush=True)
    c.close()
