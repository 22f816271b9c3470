#!/usr/bin/env python3
"""HBM traffic of the decode weight-streaming kernel from rocprofv3 PMC passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/traffic_gemv.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled;
WRITE_SIZE is exact for streaming stores.  Separate passes (FETCH_SIZE takes 3 of the 4 TCC slots).
"""
import csv
import glob
import json
import os
import sys


def per_kernel(dirname, counter, match):
    n, tot = 0, 0.0
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and match in r["Kernel_Name"]:
                n += 1
                tot += float(r["Counter_Value"])
    return n, tot


def summary(dirname, counter):
    """per-kernel average of `counter` (KiB) over the NEWEST pass found under dirname"""
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    newest = max(files, key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(newest)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {"dispatches": n, "avg_kib": t / n} for k, (n, t) in acc.items() if k.startswith(("void fl::", "fl::"))}


def main():
    if sys.argv[1] == "--summary":                # pmc_traffic.py --summary fetch_dir write_dir out.json
        json.dump({"FETCH_SIZE": summary(sys.argv[2], "FETCH_SIZE"), "WRITE_SIZE": summary(sys.argv[3], "WRITE_SIZE"),
                   "note": "raw counter averages in KiB per dispatch; gfx950: FETCH_SIZE is half the bytes of wide coalesced reads"},
                  open(sys.argv[4], "w"), indent=1)
        return
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    nf, fetch = per_kernel(fetch_dir, "FETCH_SIZE", "gemv_kernel")
    nw, write = per_kernel(write_dir, "WRITE_SIZE", "gemv_kernel")
    res = {
        "kernel": "fl::gemv_kernel (all instantiations)",
        "launches_fetch_pass": nf, "launches_write_pass": nw,
        "fetch_size_kib_per_launch_raw": fetch / max(nf, 1),
        "write_size_kib_per_launch": write / max(nw, 1),
        "read_bytes_per_launch": 2.0 * 1024.0 * fetch / max(nf, 1),        # x2: gfx950 wide-read correction
        "write_bytes_per_launch": 1024.0 * write / max(nw, 1),
        "corrections": "FETCH_SIZE x 1024 x 2 (gfx950 reports half of wide coalesced reads); WRITE_SIZE x 1024",
    }
    res["hbm_bytes_per_launch"] = res["read_bytes_per_launch"] + res["write_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
