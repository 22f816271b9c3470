#!/usr/bin/env python3
"""The prefill projection GEMM (k_gemm_8p.hip / k_gemm.hip through fl_op_linear) beside the vendor library's GEMM
(torch.nn.functional.linear = hipBLASLt) on the same layer shapes, same bf16 inputs, HIP events around 20 launches each.
A yardstick for the MFMA K loop, not a product path: the library is not linked by libfastllm_mi355x.so.

    python tools/gemm_vs_library.py
"""
import os
import sys

import numpy as np
import torch  # first: torch brings its own HIP runtime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa  # noqa: E402

SHAPES = [("mistral qkv", 512, 6144, 4096), ("mistral o", 512, 4096, 4096), ("mistral gate/up", 512, 28672, 4096),
          ("mistral down", 512, 4096, 14336), ("mistral gate/up 4k", 4096, 28672, 4096), ("mistral down 4k", 4096, 4096, 14336),
          ("qwen2 qkv 4k", 4096, 4608, 3584), ("qwen2 gate/up 4k", 4096, 37888, 3584), ("qwen2 down 4k", 4096, 3584, 18944)]


def main():
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    print("%-22s %6s %6s %6s | %10s %8s | %10s %8s | ours/lib" % ("shape", "T", "N", "K", "ours us", "TF/s", "library us", "TF/s"))
    for name, T, N, K in SHAPES:
        w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        _, ms = fa.op_linear(x, w, None, epilogue=0, iters=20)
        tw = torch.from_numpy(w.view(np.int16)).to(dev).view(torch.bfloat16)
        tx = torch.from_numpy(x.view(np.int16)).to(dev).view(torch.bfloat16)
        for _ in range(5):
            y = torch.nn.functional.linear(tx, tw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = torch.nn.functional.linear(tx, tw)
        e1.record()
        torch.cuda.synchronize()
        lib = e0.elapsed_time(e1) / 20
        fl = 2.0 * T * N * K
        print("%-22s %6d %6d %6d | %10.1f %8.1f | %10.1f %8.1f | %.2f" % (name, T, N, K, ms * 1e3, fl / ms / 1e9, lib * 1e3, fl / lib / 1e9, ms / lib), flush=True)
        del tw, tx, y
    return 0


if __name__ == "__main__":
    sys.exit(main())
