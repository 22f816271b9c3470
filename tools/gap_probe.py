"""Kernel-to-kernel gaps of one prefill from a rocprofv3 --kernel-trace csv:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/gap_probe.py run qwen2-7b 4096
  python3 tools/gap_probe.py show OUT"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np, torch, bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS[sys.argv[2]]; T = int(sys.argv[3])
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts; torch.cuda.empty_cache()
    p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 8)
    for _ in range(3):
        gm.forward_argmax(c, p, 0); c.reset(); gm.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("fl::") or "fl::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last prefill: from the last embed kernel on
    idx = max(i for i, r in enumerate(rows) if "embed" in r["Kernel_Name"])
    rows = rows[idx:]
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), rows[i]["Kernel_Name"][:40], rows[i + 1]["Kernel_Name"][:40]) for i in range(len(rows) - 1)]
    print("kernels %d  span %.3f ms  busy %.3f ms  gaps %.3f ms (mean %.2f us)" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, (t1 - t0 - busy) / 1e3 / max(1, len(gaps))))
    gaps.sort(reverse=True)
    for g in gaps[:12]:
        print("  %8.2f us  %s -> %s" % (g[0] / 1e3, g[1], g[2]))
