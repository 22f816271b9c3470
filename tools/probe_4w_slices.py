import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
for name, T, N, K in [("o 512", 512, 4096, 4096), ("down 512", 512, 4096, 14336), ("o 768", 768, 4096, 4096), ("down 768", 768, 4096, 14336), ("down 1024", 1024, 4096, 14336)]:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    line = "%-10s" % name
    for four in (0, 2):
        fa.tune("gemm_h4", 0); fa.tune("gemm_4w", four)
        _, ms = fa.op_linear(x, w, None, iters=20)
        line += "  %s slabs %6.1f us" % ("4w" if four else "8p", ms * 1e3)
    fa.tune("gemm_4w", 1); fa.tune("gemm_h4", 2); fa.tune("h4_split", 0)
    _, ms = fa.op_linear(x, w, None, iters=20)
    line += "  h4 (in-launch) %6.1f us" % (ms * 1e3)
    print(line, flush=True)
