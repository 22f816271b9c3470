"""Weight-stream rate of the T = 1 GEMV kernel vs the short-prompt GEMM (T = 2) on the decode shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa
SHAPES = [("qkv", 6144, 4096, 0), ("o_proj", 4096, 4096, 0), ("gate/up", 28672, 4096, 1), ("down", 4096, 14336, 0), ("lm_head", 32000, 4096, 0)]
rs = np.random.RandomState(0)
for name, N, K, epi in SHAPES:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for T in (1, 2, 8):
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=50)
        print("%-8s T=%d N=%6d K=%6d  %7.2f us  %7.1f GB/s" % (name, T, N, K, ms * 1e3, N * K * 2 / ms / 1e6), flush=True)
