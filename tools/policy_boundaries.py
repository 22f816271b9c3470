"""Every kernel-selection threshold of the prefill, from both sides: the default selection against the conservative one (the round-4 kernels and
rules switched off: slab sums, rope_kv_append / rmsnorm_add launches, 256-column grids, wave-pair / 16-row attention) on the same prompt --
full width, 2 layers, prefill logits + one decode step on the cache the prefill left.  usage: policy_boundaries.py [model ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
OFF = {"gemm_h4": 0, "gemm_w14": 0, "gemm_rope_4w": 0, "attn_pf32_ks2": 1, "attn_pf32_min_t": 640, "h4_nt": 0, "w14_nt": 0, "skinny_nt": 0, "rs_lazy": 0,
       "gemm_skf": 0, "prefill_dma": 0, "gateup_rowsplit": 0}   # (round 5: a rank's complete outputs on k_gemm_skf.hip, short prompts' gate/up on the ring kernel, the gate/up row split)
TS = [2, 16, 17, 32, 33, 64, 65, 128, 129, 175, 176, 191, 192, 255, 256, 257, 351, 352, 511, 512, 513, 544, 545, 576, 577, 608, 609, 639, 640, 641, 703, 704, 767, 768, 769,
      1023, 1024, 1025, 1120, 1121, 1535, 1536, 2040, 2047]
worst = 0.0
TP = int(os.environ.get("PB_TP", "1"))                  # > 1: FL_TP_EMULATED ranks (every rank's shard on this GPU: the per-rank shapes)
if os.environ.get("PB_TS"):
    TS = [int(t) for t in os.environ["PB_TS"].split(",")]
for name in sys.argv[1:] or ["mistral-7b", "qwen2-7b", "tinyllama-1.1b"]:
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=3)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16") if TP == 1 else fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", tp_mode=fa.binding.TP_EMULATED, tp_size=TP)
    del wts; torch.cuda.empty_cache()
    rs = np.random.RandomState(7)
    for T in TS:
        ids = rs.randint(0, cfg["vocab_size"], size=T + 1).astype(np.uint32)
        out = []
        for conservative in (False, True):
            fa.tune("reload_env", 0)
            if conservative:
                for k, v in OFF.items():
                    fa.tune(k, v)
            c = gm.new_cache(T + 8)
            lg = gm.forward(c, ids[:T], 0)
            out.append((lg, gm.forward(c, ids[T:T + 1], T)))
            c.close()
        fa.tune("reload_env", 0)
        rel = [float(np.linalg.norm(out[0][k] - out[1][k]) / np.linalg.norm(out[1][k])) for k in (0, 1)]
        am = [int(np.argmax(out[0][k])) == int(np.argmax(out[1][k])) for k in (0, 1)]
        worst = max(worst, *rel)
        flag = "" if max(rel) <= 1e-2 else "   <-- LARGE"
        print("%-15s T=%5d: prefill rel L2 %.2e (argmax %s)   decode %.2e (argmax %s)%s" % (name, T, rel[0], "same" if am[0] else "differs", rel[1], "same" if am[1] else "differs", flag), flush=True)
    gm.close()
print("worst rel L2 %.2e" % worst)
