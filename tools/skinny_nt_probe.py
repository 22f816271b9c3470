"""Short-prompt weight-stream kernel (k_gemm_skinny.hip): W pieces plain (skinny_nt = 1 above 32 tokens) or non-temporal (2), per projection shape,
one launch timed by events over rotating (cold) weight copies."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
SH = [("tl o", 2048, 2048, 0), ("tl qkv", 2560, 2048, 0), ("tl down", 2048, 5632, 0), ("tl gate/up", 11264, 2048, 1),
      ("qw o", 3584, 3584, 0), ("qw qkv", 4608, 3584, 0), ("mi o", 4096, 4096, 0), ("mi qkv", 6144, 4096, 0), ("mi down", 4096, 14336, 0), ("mi gate/up", 28672, 4096, 1)]
for name, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    line = "%-11s %5d x %5d (%5.1f M):" % (name, N, K, N * K / 2 ** 20)
    for T in (64, 128):
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        t = []
        for nt in (1, 2):
            fa.tune("skinny_nt", nt)
            _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=30)
            t.append(ms * 1e3)
        line += "   T=%3d plain %6.1f nt %6.1f us (x %.3f)" % (T, t[0], t[1], t[1] / t[0])
    print(line, flush=True)
fa.tune("reload_env", 0)
