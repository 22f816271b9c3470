"""Models created one after another in ONE process, each checked against the oracle (first forward after creation).
usage: seq_probe.py [rounds] [modes: comma list of tp1,emulated2,single2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
import fastllm_amd as fa
from fastllm_amd import binding
from oracle import oracle
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
modes = (sys.argv[2] if len(sys.argv) > 2 else "tp1,emulated2,single2").split(",")
KW = {"tp1": {}, "emulated2": dict(tp_mode=binding.TP_EMULATED, tp_size=2), "single2": dict(tp_mode=binding.TP_SINGLE_PROCESS, tp_size=2, device_ids=[0, 0])}
refs = {}
bad = 0
for it in range(rounds):
    for name, dtype in (("llama_a", "bf16"), ("qwen2_a", "f32"), ("mistral_a", "bf16")):
        cfg = synth.CONFIGS[name]
        w = synth.synth_weights(cfg)
        ids = synth.prompt_ids(cfg, 14, seed=11)
        if (name, dtype) not in refs:
            om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
            refs[(name, dtype)] = om.forward(om.new_cache(64), ids[:10], 0)
        o = refs[(name, dtype)]
        for mode in modes:
            g = fa.Model(cfg, w, dtype=dtype, **KW[mode])
            c = g.new_cache(64)
            a = g.forward(c, ids[:10], 0)
            d = float(np.abs(a - o).max())
            if not (d < (1e-3 if dtype == "f32" else 0.3)):
                bad += 1
                print("round %d %s %s %s: max |diff| to the oracle %.3g   <-- BAD" % (it, name, dtype, mode, d), flush=True)
            c.close(); g.close()
print("bad:", bad, "of", rounds * 3 * len(modes))
