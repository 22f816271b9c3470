#!/usr/bin/env python3
"""The roofline figure of the decode weight-streaming kernel as rocprofv3 sees it.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 64 --no-cpu-baseline
    python tools/rocprof_gemv.py <kernel_stats.csv> profiles/rNN/rocprof_gemv.json [model]

Sums every `fl::gemv_kernel<...>` instantiation of the stats file (calls, total duration), and prices the average launch with
the ALGORITHMIC bytes per launch of the benchmark's model (the decode step's weight bytes / its GEMV launches),
against the 8 TB/s HBM peak.  bench.py prints this next to its own HIP-event figure (roofline.rocprof)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastllm_amd.configs import MODEL_CONFIGS, decode_bytes_per_token   # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    model = sys.argv[3] if len(sys.argv) > 3 else "mistral-7b"
    cfg = MODEL_CONFIGS[model]
    calls, ns, rows = 0, 0.0, []
    for r in csv.DictReader(open(src)):
        if "fl::gemv_kernel<" in r["Name"]:
            calls += int(r["Calls"]); ns += float(r["TotalDurationNs"])
            rows.append({"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)})
    launches_per_step = 4 * cfg["num_hidden_layers"] + 1
    bytes_per_launch = decode_bytes_per_token(cfg, 0) / launches_per_step       # weights only (the KV cache is the attention kernel's)
    avg_us = ns / calls / 1e3
    ach = bytes_per_launch / (avg_us * 1e-6) / 1e9
    out = {"model": model, "kernel": "fl::gemv_kernel (all instantiations)", "calls": calls, "avg_launch_us": round(avg_us, 2),
           "bytes_per_launch": round(bytes_per_launch), "achieved_GBps": round(ach, 1), "peak_GBps": 8000.0, "frac": round(ach / 8000.0, 4),
           "source": os.path.basename(src), "instantiations": rows,
           "note": "rocprofv3 --kernel-trace --stats over bench.py (decode, prefill and parity-gate launches of the kernel included)"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
