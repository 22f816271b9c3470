#!/usr/bin/env python3
"""Summarise FL_8P_STAMPS records (k_gemm_8p.hip diagnostic instantiation): per-workgroup prologue / K loop / epilogue time,
core clock held in the loop, start skew and occupancy per XCD.  Usage: python tools/stamps_8p.py file [launch index]"""
import sys
import numpy as np

def main():
    recs, cur = [], None
    for ln in open(sys.argv[1]):
        if ln.startswith("launch"):
            cur = (ln.split()[1:], [])
            recs.append(cur)
        else:
            cur[1].append([int(v) for v in ln.split()])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    hdr, rows = recs[which]
    a = np.array(rows, dtype=np.float64)
    t0 = a[:, 0].min()
    rt = (a[:, [0, 2, 4, 6]] - t0) / 100.0          # us
    clk = (a[:, [3, 5, 7]] - a[:, [1, 3, 5]]) / np.maximum(1, (a[:, [2, 4, 6]] - a[:, [0, 2, 4]])) * 100.0   # MHz
    pro, loop, epi = rt[:, 1] - rt[:, 0], rt[:, 2] - rt[:, 1], rt[:, 3] - rt[:, 2]
    print("launch T,N,K,epi,nwg =", hdr, " wall (first entry -> last exit) %.1f us" % rt[:, 3].max())
    for name, v in (("entry", rt[:, 0]), ("prologue", pro), ("k loop", loop), ("epilogue", epi), ("exit", rt[:, 3])):
        print("  %-9s min %8.2f  median %8.2f  p90 %8.2f  max %8.2f us" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    print("  core clock: prologue %.0f  loop %.0f  epilogue %.0f MHz (median)" % tuple(np.median(clk, axis=0)))
    xcc = a[:, 9].astype(int) & 15
    cu = (a[:, 8].astype(int) >> 8) & 15
    se = (a[:, 8].astype(int) >> 13) & 7
    key = xcc * 1000 + se * 16 + cu
    per_cu = {}
    for k, e, x in zip(key, rt[:, 0], rt[:, 3]):
        per_cu.setdefault(k, []).append((e, x))
    n = np.array([len(v) for v in per_cu.values()])
    print("  distinct CUs %d; tiles per CU min %d max %d" % (len(per_cu), n.min(), n.max()))
    gaps = []
    for v in per_cu.values():
        v.sort()
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    if gaps:
        g = np.array(gaps)
        print("  gap between a tile's exit and the next tile's entry on the same CU: median %.2f  p90 %.2f  max %.2f us" % (np.median(g), np.percentile(g, 90), g.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  xcc %d: %4d tiles, loop median %.2f us, last exit %.1f us" % (x, m.sum(), np.median(loop[m]), rt[m, 3].max()))

if __name__ == "__main__":
    main()
