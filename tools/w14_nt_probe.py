"""gemm_w14 with plain / non-temporal W pieces on Mistral-7B's gate/up matrix, one launch timed by events over rotating (cold) weight copies."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
N, K = 28672, 4096
w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
for T in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    line = "gate/up 28672 x 4096, T=%4d:" % T
    for nt in (0, 1):
        fa.tune("gemm_h4", 0); fa.tune("gemm_w14", 2); fa.tune("w14_nt", nt)
        _, ms = fa.op_linear(x, w, None, epilogue=1, iters=20)
        line += "   nt=%d %7.1f us  %6.0f TFLOP/s" % (nt, ms * 1e3, 2.0 * T * N * K / ms / 1e9)
    print(line, flush=True)
    fa.tune("reload_env", 0)
