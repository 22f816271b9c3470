"""Worker of tests/test_gpu_h4_shared_gpu.py: K-sliced 128 x 256 GEMMs on integer operands, over and over, while other processes do the
same on the same GPU; every result must be the exact integer product.  argv: seed, rounds, wait_us"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastllm_amd as fa
import synth

seed, rounds, wait_us = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = np.random.RandomState(seed)
cases = []
for (T, N, K, ks) in [(512, 4096, 4096, 4), (512, 4096, 2048, 2), (384, 2048, 6144, 3)]:
    x = rs.randint(-3, 4, size=(T, K)).astype(np.float32)
    w = rs.randint(-3, 4, size=(N, K)).astype(np.float32)
    cases.append((synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w), (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32), ks))
fa.tune("gemm_h4", 2); fa.tune("h4_wait_us", wait_us)
bad = 0
for r in range(rounds):
    xb, wb, ref, ks = cases[r % len(cases)]
    fa.tune("h4_split", ks)
    y = fa.op_linear(xb, wb, None)
    bad += int(not np.array_equal(y, ref))
print("worker %d: %d rounds, %d wrong" % (seed, rounds, bad), flush=True)
sys.exit(1 if bad else 0)
