"""What would Infinity-Cache-resident weights buy each of Mistral-7B's four projections at T tokens?  One launch timed by events, weights
cold (rotating copies, > 256 MiB together) against one copy replayed (op_hot = 1: resident in the Infinity Cache, far beyond the L2s)."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fastllm_amd as fa
rs = np.random.RandomState(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SH = [("qkv", 6144, 4096, 0), ("o_proj", 4096, 4096, 0), ("gate/up", 28672, 4096, 1), ("down", 4096, 14336, 0)]
tot = [0.0, 0.0]
for name, N, K, epi in SH:
    w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
    line = "%-8s %5d x %5d T=%d:" % (name, N, K, T)
    for i, hot in enumerate((0, 1)):
        fa.tune("op_hot", hot)
        _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=24)
        tot[i] += ms * 1e3
        line += "   %s %6.1f us" % ("resident" if hot else "cold    ", ms * 1e3)
    print(line, flush=True)
fa.tune("reload_env", 0)
print("sum: cold %.1f us, resident %.1f us per layer" % tuple(tot))
