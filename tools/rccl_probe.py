import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
if os.environ.get("WITH_TORCH") == "1":
    import torch
    print("torch", torch.__version__, torch.cuda.is_available(), flush=True)
import numpy as np
import fastllm_amd as fa
import synth
cfg = synth.CONFIGS["llama_a"]
w = synth.synth_weights(cfg)
print("creating", flush=True)
m = fa.Model(cfg, w, dtype="bf16")
print("created", flush=True)
c = m.new_cache(64)
ids = synth.prompt_ids(cfg, 8)
print("prefill", m.forward_argmax(c, ids, 0), flush=True)
print("decode", m.decode_greedy(c, 3, 8, 6), flush=True)
print("done", flush=True)
