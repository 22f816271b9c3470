"""One token past a tile edge: Mistral-7B's projections at T = 513 ... on the candidate kernels, one launch between event pairs over cold weight copies.
usage: cliff_probe.py [T ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastllm_amd as fa
rs = np.random.RandomState(0)
SH = [("gate/up", 28672, 4096, 1), ("down", 4096, 14336, 0), ("qkv", 6144, 4096, 0), ("o_proj", 4096, 4096, 0)]
Ts = [int(a) for a in sys.argv[1:]] or [512, 513, 576, 640, 700, 768]
for name, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for T in Ts:
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        line = "%-8s T=%4d:" % (name, T)
        for label, sw in (("default", {}), ("h4/1", {"gemm_h4": 2, "h4_split": 1}), ("h4/2", {"gemm_h4": 2, "h4_split": 2}), ("h4/4", {"gemm_h4": 2, "h4_split": 4}),
                          ("no-w14-no-h4", {"gemm_w14": 0, "gemm_h4": 0})):
            if epi == 1 and label in ("h4/2", "h4/4"):
                continue
            try:
                for k, v in sw.items():
                    fa.tune(k, v)
                fa.tune("op_maxsplit", 8)
                _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=16)
                line += "  %s %6.1f us" % (label, ms * 1e3)
            except Exception as e:
                line += "  %s n/a" % label
            finally:
                fa.tune("reload_env", 0)
        print(line, flush=True)
