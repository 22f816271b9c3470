"""GEMV time for row-parallel tensor-parallel slices whose K ends in a partial 64-lane chunk."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastllm_amd as fa
rs = np.random.RandomState(0)
for N, K in ((4096, 1792), (4096, 2048), (3584, 4736), (3584, 4608), (3584, 896), (4096, 512)):
    w = rs.randint(0, 1 << 15, size=(N, K)).astype(np.uint16)
    x = rs.randint(0, 1 << 15, size=(1, K)).astype(np.uint16)
    y, ms = fa.op_linear(x, w, iters=200)
    print("N=%5d K=%5d (chunks %% 64 = %2d): %6.2f us  %6.1f GB/s" % (N, K, (K // 8) % 64, ms * 1e3, N * K * 2 / ms / 1e6), flush=True)
