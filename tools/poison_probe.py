"""FL_DEBUG_POISON=1: which small-model configuration reads bytes nobody wrote?  Every mode against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FL_DEBUG_POISON", "255")
import numpy as np
import synth
import fastllm_amd as fa
from fastllm_amd import binding
from oracle import oracle
names = sys.argv[1:] or ["llama_a", "qwen2_a", "mistral_a", "llama_mha", "llama_tp4", "llama_d100", "qwen2_d96", "mistral_d48"]
for name in names:
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, 14, seed=11)
    for dtype in ("bf16", "f32"):
        om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
        oc = om.new_cache(64)
        o = [om.forward(oc, ids[:10], 0)] + [om.forward(oc, ids[i:i + 1], i) for i in range(10, 13)]
        for mode, kw in (("tp1", {}), ("emulated2", dict(tp_mode=binding.TP_EMULATED, tp_size=2)), ("single2", dict(tp_mode=binding.TP_SINGLE_PROCESS, tp_size=2, device_ids=[0, 0]))):
            if mode != "tp1" and (cfg.get("num_key_value_heads") or cfg["num_attention_heads"]) % 2:
                continue
            try:
                g = fa.Model(cfg, w, dtype=dtype, **kw)
                c = g.new_cache(64)
                got = [g.forward(c, ids[:10], 0)] + [g.forward(c, ids[i:i + 1], i) for i in range(10, 13)]
            except fa.FastLLMError as e:
                print("%-12s %-5s %-10s ERROR %s   <-- BAD" % (name, dtype, mode, str(e)[:90]), flush=True)
                continue
            d = [float(np.abs(a - b).max()) if np.isfinite(a).all() else float("nan") for a, b in zip(got, o)]
            flag = "" if all(x == x and x < (1e-3 if dtype == "f32" else 0.3) for x in d) else "   <-- BAD"
            print("%-12s %-5s %-10s prefill %.3g decode %s%s" % (name, dtype, mode, d[0], " ".join("%.3g" % x for x in d[1:]), flag), flush=True)
            c.close(); g.close()
