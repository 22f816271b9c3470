import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
SH = [("mistral o", 512, 4096, 4096, 0), ("mistral down", 512, 4096, 14336, 0), ("mistral qkv", 512, 6144, 4096, 0)]
for name, T, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for ks in (1, 2, 4):
        line = "%-14s h4/%d hot=%s:" % (name, ks, os.environ.get("FL_OP_HOT", "0"))
        for pf in (6, 65536 + 6):
            fa.tune("gemm_h4", 2); fa.tune("h4_split", ks); fa.tune("h4_pf", pf)
            _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=20)
            line += "  %s %6.1f" % ("blocked" if pf >> 16 else "rowmajor", ms * 1e3)
        print(line, flush=True)
