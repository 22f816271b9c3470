#!/usr/bin/env python3
"""Time the prefill projection GEMM on the Mistral-7B / Qwen2-7B layer shapes (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa

SHAPES = [("mistral qkv", 512, 6144, 4096, 0), ("mistral o", 512, 4096, 4096, 0), ("mistral gate/up", 512, 28672, 4096, 1),
          ("mistral down", 512, 4096, 14336, 0), ("qwen2 qkv 4k", 4096, 4608, 3584, 0), ("qwen2 gate/up 4k", 4096, 37888, 3584, 1),
          ("qwen2 down 4k", 4096, 3584, 18944, 0), ("tinyllama gate/up 128", 128, 11264, 2048, 1)]
rs = np.random.RandomState(0)
for name, T, N, K, epi in SHAPES:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=20)
    print("%-24s T=%5d N=%6d K=%6d  %9.1f us  %7.1f TFLOP/s  (weights at %6.1f GB/s)" % (name, T, N, K, ms * 1e3, 2.0 * T * N * K / ms / 1e9, N * K * 2 / ms / 1e6), flush=True)
