"""Probe of a flaky test: FL_TP_SINGLE_PROCESS with both shards on ONE device against FL_TP_EMULATED, model after model in one process.
usage: sp_probe.py [iterations] [decode steps after the prefill (0: prefill only)]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
import fastllm_amd as fa
from fastllm_amd import binding
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ndec = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for it in range(iters):
    for name, dtype in (("llama_a", "bf16"), ("qwen2_a", "f32"), ("mistral_a", "bf16")):
        cfg = synth.CONFIGS[name]
        w = synth.synth_weights(cfg)
        gS = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_SINGLE_PROCESS, tp_size=2, device_ids=[0, 0])
        gE = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=2)
        ids = synth.prompt_ids(cfg, 14, seed=11)
        cS, cE = gS.new_cache(64), gE.new_cache(64)
        a, b = gS.forward(cS, ids[:10], 0), gE.forward(cE, ids[:10], 0)
        if not np.array_equal(a, b):
            bad += 1
            cS2 = gS.new_cache(64)
            a2 = gS.forward(cS2, ids[:10], 0)
            from oracle import oracle
            om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
            o = om.forward(om.new_cache(64), ids[:10], 0)
            print(it, name, "PREFILL MISMATCH", float(np.abs(a - b).max()), "| again: single==emulated", np.array_equal(a2, b),
                  "| vs oracle: single %.3g emulated %.3g" % (float(np.abs(a - o).max()), float(np.abs(b - o).max())), flush=True)
        if ndec:
            for i in range(10, 14):
                x, y = gS.forward(cS, ids[i:i + 1], i), gE.forward(cE, ids[i:i + 1], i)
                if not np.array_equal(x, y):
                    bad += 1; print(it, name, "decode mismatch at", i, float(np.abs(x - y).max()), flush=True)
            f = gS.forward_argmax(cS, ids[:1], 14); gE.forward_argmax(cE, ids[:1], 14)
            if not np.array_equal(gS.decode_greedy(cS, f, 15, ndec), gE.decode_greedy(cE, f, 15, ndec)):
                bad += 1; print(it, name, "greedy mismatch", flush=True)
        gS.close(); gE.close()
        del cS, cE, gS, gE
print("mismatches:", bad, "(iterations %d, decode %d, FL_TP_GRAPH=%s)" % (iters, ndec, os.environ.get("FL_TP_GRAPH", "1")))
