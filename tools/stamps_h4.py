"""Read FL_H4_STAMPS records (k_gemm_h4.hip): per launch, microseconds from the earliest workgroup start to each stamp, median | max over
the workgroups.  Stamps: 0 start, 1 prologue done, 2 K loop done, 3 published, 4 all slices there (thread 0), 5 claimed + barrier,
6 first block summed, 7 own blocks stored."""
import sys
import numpy as np
names = ["start", "prologue", "kloop", "published", "all-there", "claimed", "summed", "stored"]
lines = open(sys.argv[1]).read().split("\n")
i = 0
while i < len(lines):
    if not lines[i].startswith("launch"):
        i += 1; continue
    _, T, N, K, epi, ks, nwg = lines[i].split()
    nwg = int(nwg)
    a = np.array([[int(x) for x in lines[i + 1 + j].split()] for j in range(nwg)], dtype=np.float64)
    i += 1 + nwg
    t0 = a[:, 0].min()
    us = (a - t0) / 100.0
    us[a == 0] = np.nan
    print("T=%s N=%s K=%s epi=%s slices=%s (%d workgroups)" % (T, N, K, epi, ks, nwg))
    for j in range(8):
        col = us[:, j]
        if np.all(np.isnan(col)):
            continue
        print("   %-10s median %7.2f   min %7.2f   max %7.2f" % (names[j], np.nanmedian(col), np.nanmin(col), np.nanmax(col)))
