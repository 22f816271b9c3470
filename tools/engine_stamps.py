#!/usr/bin/env python3
"""Reads the file FL_ENGINE_STAMPS=<file> leaves (k_engine.hip: one block per engine launch, one line of 32 wall-clock stamps
in 10-ns ticks per workgroup, relative to the workgroup's own start) and prints, per op of the chain, the median / max over
the workgroups of: when the CU's streamers were done with the previous op, when the op's input was complete in LDS (gather
done), when streamer wave 0 was past the barrier, when it had finished its rows.

    FL_ENGINE_STAMPS=/tmp/s.txt FL_GRAPH=0 python tools/engine_probe.py ... ; python tools/engine_stamps.py /tmp/s.txt [launch index]
"""
import sys

import numpy as np


def main():
    path = sys.argv[1]
    pick = int(sys.argv[2]) if len(sys.argv) > 2 else None
    launches, cur = [], None
    for ln in open(path):
        if ln.startswith("launch"):
            cur = [ln.strip(), []]
            launches.append(cur)
        elif ln.startswith("wg") and cur is not None:
            cur[1].append([int(x) for x in ln.split()[2:]])
    if pick is None:
        pick = len(launches) // 2
    # aggregate over launches with the same tag as `pick`
    tag = launches[pick][0]
    same = [np.array(l[1], dtype=np.float64) for l in launches if l[0] == tag and len(l[1]) == len(launches[pick][1])]
    arr = np.stack(same[len(same) // 4:]) / 100.0            # us; drop the first quarter (warm-up)
    arr[arr < 0] = np.nan
    print(tag, "-- %d launches, %d workgroups; us from each workgroup's own start (median over launches of median | max over workgroups)" % (arr.shape[0], arr.shape[1]))
    names = {4: "local streamers done with prev op", 1: "input complete (gather done)", 3: "streamer 0 past barrier", 2: "streamer 0 rows done"}
    nops = int(tag.split("nops")[1].split()[0])
    for o in range(nops):
        for slot in (4, 1, 3, 2):
            col = arr[:, :, slot + 4 * o]
            if np.all(np.isnan(col)):
                continue
            print("  op %d  %-36s %7.2f | %7.2f" % (o, names[slot], np.nanmedian(np.nanmedian(col, axis=1)), np.nanmedian(np.nanmax(col, axis=1))))
    for o in range(nops):
        col = arr[:, :, 16 + o]
        if not np.all(np.isnan(col)):
            print("  op %d  %-36s %7.2f | %7.2f" % (o, "loader 0 requests the op's first block", np.nanmedian(np.nanmedian(col, axis=1)), np.nanmedian(np.nanmax(col, axis=1))))
    for slot, what in ((20, "loader 0: all requested"), (21, "loader 0: all landed"), (22, "loader 0: us blocked on a full ring"), (23, "loader 0: us in thin mode waits")):
        col = arr[:, :, slot]
        if not np.all(np.isnan(col)):
            print("        %-36s %7.2f | %7.2f" % (what, np.nanmedian(np.nanmedian(col, axis=1)), np.nanmedian(np.nanmax(col, axis=1))))
    end = arr[:, :, 31]
    print("  end   %-36s %7.2f | %7.2f" % ("workgroup exit", np.nanmedian(np.nanmedian(end, axis=1)), np.nanmedian(np.nanmax(end, axis=1))))


if __name__ == "__main__":
    main()
