#!/usr/bin/env python3
"""Sweep the decode GEMV kernel's tuning knobs on the Mistral-7B / TinyLlama layer shapes (cold caches: fl_op_linear
rotates over copies of W that exceed the Infinity Cache; round 1's sweep re-read ONE copy and flattered the short matrices).
Usage (GPU box): python tools/gemv_sweep.py > gpurun_out/gemv_sweep.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa

SHAPES = [("mistral qkv", 6144, 4096, 0), ("mistral o", 4096, 4096, 0), ("mistral gate/up", 28672, 4096, 1),
          ("mistral down", 4096, 14336, 0), ("mistral lm_head", 32000, 4096, 0),
          ("tinyllama qkv", 2560, 2048, 0), ("tinyllama gate/up", 11264, 2048, 1), ("tinyllama down", 2048, 5632, 0)]


def main():
    rs = np.random.RandomState(0)
    for name, N, K, epi in SHAPES:
        w = rs.randint(0, 65536, size=(N, K), dtype=np.uint16) & 0xBFFF   # finite bf16 bit patterns
        w = (w & 0x807F) | 0x3C00                                        # |w| in [1/64.. ) small, finite
        x = (rs.randint(0, 65536, size=(1, K), dtype=np.uint16) & 0x807F) | 0x3C00
        best = None
        for R, U in ((2, 2), (2, 4), (4, 2)):
            for blocks, waves in ((0, 0), (256, 6), (256, 8), (256, 10), (256, 12), (512, 4), (512, 6), (768, 4), (1024, 4), (128, 12)):
                fa.tune("gemv_r", R); fa.tune("gemv_u", U); fa.tune("gemv_blocks", blocks); fa.tune("gemv_waves", waves)
                _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=50)
                gbs = N * K * 2 / ms / 1e6
                print("%-18s N=%6d K=%6d R=%d U=%d blocks=%4d waves=%2d  %8.2f us  %7.1f GB/s" % (name, N, K, R, U, blocks, waves, ms * 1e3, gbs), flush=True)
                if best is None or gbs > best[0]:
                    best = (gbs, R, U, blocks, waves)
        print("BEST %-18s %7.1f GB/s R=%d U=%d blocks=%d waves=%d" % ((name,) + best), flush=True)


if __name__ == "__main__":
    main()
