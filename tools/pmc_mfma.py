#!/usr/bin/env python3
"""MFMA utilisation of the prefill kernels from a rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py ...
    python tools/pmc_mfma.py DIR out.json

rocprofv3 reports each counter summed over the 8 XCDs.  MfmaUtil (ROCm's derived metric) =
sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max-over-XCD(GRBM_GUI_ACTIVE) * SIMD count); with the per-XCD split not in the CSV,
max(GRBM_GUI_ACTIVE) is taken as sum / 8 (all XCDs are busy for the whole kernel on these grids); 1024 SIMDs.
"""
import collections, csv, glob, json, os, sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    newest = max(files, key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(newest)):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm" not in k and "attn_prefill" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
    res = {}
    for k, v in acc.items():
        ga, mf = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if ga:
            res[k] = {"dispatches": n[k], "mfma_busy_cycles_per_dispatch": mf / n[k], "gui_active_sum_per_dispatch": ga / n[k],
                      "MfmaUtil_percent": round(100.0 * mf / ((ga / 8.0) * 1024.0), 1)}
    json.dump({"formula": "100 * SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs)", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
