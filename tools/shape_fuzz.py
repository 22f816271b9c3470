"""tests/test_gpu_ops.py's seeded projection-shape fuzz over more seeds than the suite carries.  usage: shape_fuzz.py first_seed last_seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fastllm_amd as fa
import synth
import test_gpu_ops as t
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b + 1):
    for T, N, K, epi, bias in t._fuzz_shapes(seed, 14):
        x, w = t._rand((T, K), 1000 + seed), t._rand((N, K), 2000 + seed, 0.05)
        bb = t._rand((N,), 3000 + seed) if bias else None
        for dtype in ("bf16", "f32"):
            if dtype == "f32" and T * N * K > 3e9:
                continue
            try:
                if dtype == "bf16":
                    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
                    ref = t._ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), bb, epi)
                    y = fa.op_linear(xb, wb, bb, epilogue=epi)
                else:
                    ref = t._ref(x, w, bb, epi)
                    y = fa.op_linear(x, w, bb, epilogue=epi)
                if epi and dtype == "bf16":
                    ok = np.allclose(y, ref, atol=2e-3, rtol=2 ** -7)
                else:
                    ok = np.allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)
            except Exception as e:
                ok = False
                print("  exception:", e)
            if not ok:
                bad += 1
                print("MISMATCH seed %d %s T=%d N=%d K=%d epi=%d bias=%s" % (seed, dtype, T, N, K, epi, bias), flush=True)
    print("seed %d done" % seed, flush=True)
print("shapes checked over seeds %d..%d, mismatches: %d" % (a, b, bad))
