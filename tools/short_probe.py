"""Short prompts (T <= 128): each of Mistral-7B's projections on the weight-stream kernel (k_gemm_skinny.hip) against the 128 x 256 kernel
with in-launch K slices (gemm_h4 = 2), one launch timed by events over rotating (cold) weight copies."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
SH = [("gate/up", 28672, 4096, 1), ("qkv", 6144, 4096, 0), ("o_proj", 4096, 4096, 0), ("down", 4096, 14336, 0)]
Ts = [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128]
for name, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for T in Ts:
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        line = "%-8s T=%4d:" % (name, T)
        fa.tune("gemm_h4", 0)
        _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=24)
        line += "  default %6.1f us (%.2f TB/s)" % (ms * 1e3, N * K * 2 / ms / 1e9)
        for ks in (1, 2, 4):
            fa.tune("gemm_h4", 2); fa.tune("h4_split", ks)
            try:
                _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=24)
                line += "  h4/%d %6.1f" % (ks, ms * 1e3)
            except Exception as e:
                line += "  h4/%d   n/a " % ks
        print(line, flush=True)
        fa.tune("reload_env", 0)
