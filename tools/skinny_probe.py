#!/usr/bin/env python3
"""Weight-stream rate of the short-prompt / batched-decode projection GEMM (k_gemm_skinny.hip) on the Mistral-7B and
TinyLlama layer shapes, cold caches (fl_op_linear rotates over copies of W).  FL_SKINNY_STAGES / FL_SKINNY_NT pick the ring."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa

SH = [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate/up", 28672, 4096, 1), ("down", 4096, 14336, 0), ("lm_head", 32000, 4096, 0),
      ("tl qkv", 2560, 2048, 0), ("tl gate/up", 11264, 2048, 1), ("tl down", 2048, 5632, 0)]
Ts = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1,8,32,128").split(",")]
rs = np.random.RandomState(0)
for name, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for T in Ts:
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=48)
        print("%-10s T=%4d N=%6d K=%6d  %8.2f us  weights at %7.1f GB/s" % (name, T, N, K, ms * 1e3, N * K * 2 / ms / 1e6), flush=True)
