"""Per-token cost of device-side temperature sampling vs ArgMax (TinyLlama V=32000, Qwen2 V=152064 logits)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fastllm_amd as fa

for V in (32000, 152064):
    lg = (np.random.RandomState(0).randn(V) * 3).astype(np.float32)
    for temp in (0.0, 0.8):
        n = 2000
        fa.op_sample(lg, 10, temp)
        t0 = time.perf_counter()
        fa.op_sample(lg, n, temp)
        dt = time.perf_counter() - t0
        print("V=%6d temperature=%.1f: %.1f us per selection (launch included)" % (V, temp, dt / n * 1e6))
