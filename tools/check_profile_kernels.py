#!/usr/bin/env python3
"""Does a rocprofv3 kernel-stats file describe the kernels a bench line / profile listing says ran?

    python tools/check_profile_kernels.py <kernel_stats.csv> <bench line .json | prefill_profile .txt> [--stamp]

VERDICT r4: a committed stats file showed attn_prefill_mfma_kernel where the bench by then launched attn_prefill32_kernel.  Every
kernel CLASS named in the bench line's `prefill.kernels` / `kernels` (or in a tools/prefill_profile.py listing) is mapped to the HIP
symbol that implements it; the script fails (exit 1) when a symbol is missing from the stats file.  --stamp writes <stats>.commit.txt
with the commit the GPU box ran (tools/gpu.sh leaves it in .fl_commit)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# class label (fl_profile tag) -> substring of the kernel symbol
RULES = [(r"^gemm_mfma\[h4,", "gemm_h4_kernel"), (r"^gemm_mfma\[w14,", "gemm_w14_kernel"), (r"^gemm_mfma\[8p,fixup", "gemm_8p_fixup_kernel"),
         (r"^gemm_mfma\[8p,", "gemm_8p_kernel"), (r"^gemm_mfma\[4w,", "gemm_4w_kernel"), (r"^gemm_mfma\[skinny,", "gemm_skinny_kernel"),
         (r"^gemm_mfma\[skf,", "gemm_skf_kernel"), (r"^gemm_mfma\[256x128", "gemm_mfma256_kernel"), (r"^gemm_mfma\[128x128", "gemm_mfma_kernel"),
         (r"^gemm_mfma\[f32", "gemm_f32_mfma_kernel"), (r"^attn_prefill\[32row", "attn_prefill32_kernel"), (r"^attn_prefill$", "attn_prefill_mfma_kernel"),
         (r"^attn_decode", "attn_decode_mfma"), (r"^rmsnorm_add\[finalize", "rms_finalize_kernel"), (r"^rmsnorm_add", "rmsnorm_add_kernel"),
         (r"^rope_kv", "rope_kv"), (r"^gemv\[b\d+d:", "gemv_dma_kernel"), (r"^gemv\[b\d+:", "gemv_batch"), (r"^gemv\[", "gemv_kernel"),
         (r"^select_advance", "select_advance"), (r"^embed", "embed"), (r"^comm_oneshot", "oneshot_kernel")]


def classes_of(path):
    text = open(path).read()
    if path.endswith(".json"):
        try:
            j = json.loads(text)
        except ValueError:
            j = json.loads(text.strip().splitlines()[-1])          # (a bench log: the JSON line is the last)
        j = j.get("parsed") or j                                    # (a driver record, BENCH_rNN.json, wraps the line)
        names = [k["name"] for k in (j.get("prefill") or {}).get("kernels", [])] + [k["name"] for k in j.get("kernels", [])]
    else:
        names = re.findall(r"^\s+(\S+)\s+x\d+", text, re.M)
    return sorted(set(names))


def main():
    stats, src = sys.argv[1], sys.argv[2]
    syms = [r["Name"] for r in csv.DictReader(open(stats))]
    missing, unknown = [], []
    classes = classes_of(src)
    if not classes:
        print("no kernel classes found in %s (a bench line's prefill.kernels / kernels, or a prefill_profile listing)" % src)
        raise SystemExit(2)
    for c in classes:
        for pat, sym in RULES:
            if re.search(pat, c):
                if not any(sym in s for s in syms):
                    missing.append((c, sym))
                break
        else:
            unknown.append(c)
    if "--stamp" in sys.argv:
        cf = os.path.join(ROOT, ".fl_commit")
        with open(stats + ".commit.txt", "w") as f:
            f.write("%s\ncollected for: %s\n" % (open(cf).read().strip() if os.path.exists(cf) else "unknown", os.path.basename(src)))
    for c in unknown:
        print("note: no symbol rule for class %s" % c)
    for c, s in missing:
        print("MISSING: the line names %s but no kernel %s is in %s" % (c, s, os.path.basename(stats)))
    print("%s: %d kernel symbols; %s" % (os.path.basename(stats), len(syms), "all of the line's kernels present" if not missing else "%d missing" % len(missing)))
    raise SystemExit(1 if missing else 0)


if __name__ == "__main__":
    main()
