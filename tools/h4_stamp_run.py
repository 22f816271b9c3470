import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
for (T, N, K, ks) in [(512, 4096, 4096, 4), (512, 4096, 14336, 4), (512, 4096, 4096, 2)]:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    fa.tune("gemm_h4", 2); fa.tune("h4_split", ks)
    for _ in range(3):
        fa.op_linear(x, w, None)
