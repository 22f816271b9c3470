"""Mid-size prompts (129-512 tokens): the narrow projections (o_proj, QKV) on the short-prompt kernel run as ceil(T / 128) token blocks per
strip (FL_GEMM_SKINNY_MAXT) against the default plan.  One launch between event pairs over cold weight copies (fl_op_linear).
usage: skinny_mid_probe.py [T ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastllm_amd as fa
rs = np.random.RandomState(0)
SH = [("o_proj", 4096, 4096), ("qkv", 6144, 4096), ("down", 4096, 14336)]
Ts = [int(a) for a in sys.argv[1:]] or [200, 256, 384, 512]
for name, N, K in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    for T in Ts:
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        line = "%-8s T=%4d:" % (name, T)
        for label, sw in (("default", {"op_maxsplit": 8}),
                          ("skinny/1", {"gemm_h4": 0, "gemm_w14": 0, "gemm_skinny_maxt": 1024, "op_maxsplit": 1}),
                          ("skinny/2", {"gemm_h4": 0, "gemm_w14": 0, "gemm_skinny_maxt": 1024, "op_maxsplit": 2}),
                          ("skinny/4", {"gemm_h4": 0, "gemm_w14": 0, "gemm_skinny_maxt": 1024, "op_maxsplit": 4})):
            try:
                for k, v in sw.items():
                    fa.tune(k, v)
                _, ms = fa.op_linear(x, w, None, epilogue=0, iters=16)
                line += "  %s %6.1f us" % (label, ms * 1e3)
            except Exception as e:
                line += "  %s n/a (%s)" % (label, str(e)[:40])
            finally:
                fa.tune("reload_env", 0)
        print(line, flush=True)
