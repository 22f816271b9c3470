#!/usr/bin/env python3
"""Summarise FL_SKINNY_STAMPS records (k_gemm_skinny.hip diagnostic instantiations; FL_SKINNY_LOADERS=0: the shipped kernel at 128 tokens, default: the loader-wave variant): core-clock cycles
per K step a wave spends in the counted vmcnt wait / at the barrier / issuing LDS-DMA / in fragment reads + MFMAs.
Usage: FL_SKINNY_STAMPS=f python tools/stamps_run.py 128 N K epi; python tools/stamps_skinny.py f"""
import sys
import numpy as np

recs, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("launch"):
        cur = (ln.split()[1:], [])
        recs.append(cur)
    else:
        cur[1].append([int(v) for v in ln.split()])
hdr, rows = recs[-1]
a = np.array(rows, dtype=np.float64)
nwv = int(hdr[5])
nk = a[:, 5]
per = a[:, :4] / nk[:, None]
print("launch T,N,K,epi =", hdr[:4], " waves", len(a), " K steps per wave %.0f" % nk.mean())
names = ("vmcnt wait", "barrier", "dma issue", "reads+mfma")
for i, n in enumerate(names):
    print("  %-11s median %7.0f  p10 %7.0f  p90 %7.0f cycles per K step" % (n, np.median(per[:, i]), np.percentile(per[:, i], 10), np.percentile(per[:, i], 90)))
print("  loop total  median %7.0f cycles per K step" % np.median(a[:, 4] / nk))
w = np.arange(len(a)) % nwv
for k in range(nwv):
    m = w == k
    print("  wave %d: wait %6.0f  barrier %6.0f  issue %6.0f  compute %6.0f" % ((k,) + tuple(np.median(per[m, i]) for i in range(4))))
