#!/usr/bin/env python3
"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected (one or more DIRs), as JSON.
    python tools/pmc_table.py out.json DIR [DIR ...]        (kernels whose name contains gemm / attn_prefill / rmsnorm / rope)
Derived where the inputs are there: MfmaUtil % = 100 * SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs);
lds_insts_per_mfma_busy_kcycle; bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (or / SQ_ACTIVE_INST_LDS)."""
import collections, csv, glob, json, os, sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in ("gemm", "attn_prefill", "rmsnorm", "rope", "rms_finalize")):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    res = {}
    for k, v in acc.items():
        e = {"dispatches": max(cnt[k].values())}
        for c, x in sorted(v.items()):
            e[c + "_per_dispatch"] = round(x / cnt[k][c], 1)
        ga, mf = v.get("GRBM_GUI_ACTIVE"), v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if ga and mf is not None:
            e["MfmaUtil_percent"] = round(100.0 * (mf / cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / ((ga / cnt[k]["GRBM_GUI_ACTIVE"] / 8.0) * 1024.0), 1)
        if v.get("SQ_LDS_BANK_CONFLICT") is not None:
            den = v.get("SQ_LDS_IDX_ACTIVE") or v.get("SQ_ACTIVE_INST_LDS")
            if den:
                e["lds_bank_conflict_share"] = round(v["SQ_LDS_BANK_CONFLICT"] / den, 4)
        if v.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if v.get(c) is not None:
                    e[c + "_share_of_wave_cycles"] = round(v[c] / cnt[k][c] / (v["SQ_WAVE_CYCLES"] / cnt[k]["SQ_WAVE_CYCLES"]), 4)
        res[k] = e
    json.dump({"notes": "sums over the 8 XCDs as rocprofv3 reports them, divided by the dispatch count; SQ_* wave-state counters are in quad-cycles",
               "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
