"""The 256 x 224 kernel (gemm_w14) against the 256 x 256 one (gemm_4w) on Mistral-7B's gate/up matrix, one launch timed by events over a
rotating set of weight copies (cold in the Infinity Cache): us per launch and the TFLOP/s of the tile grid."""
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
    for mode in (0, 2):
        fa.tune("gemm_h4", 0); fa.tune("gemm_w14", mode)
        _, ms = fa.op_linear(x, w, None, epilogue=1, iters=20)
        line += "   w14=%d %7.1f us  %6.0f TFLOP/s" % (mode, ms * 1e3, 2.0 * T * N * K / ms / 1e9)
    print(line, flush=True)
    fa.tune("reload_env", 0)
