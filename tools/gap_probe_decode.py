"""Kernel-to-kernel gaps of graph-replayed decode steps from a rocprofv3 --kernel-trace csv:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/gap_probe_decode.py run mistral-7b
  python3 tools/gap_probe_decode.py show OUT"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np, torch, bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS[sys.argv[2]]; T = 512
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts; torch.cuda.empty_cache()
    p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    c = gm.new_cache(T + 200)
    f = gm.forward_argmax(c, p, 0)
    gm.decode_greedy(c, f, T, 96)
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "fl::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = [i for i, r in enumerate(rows) if "select_advance" in r["Kernel_Name"]]
    a, b = sel[-20], sel[-4]                      # 16 steady-state steps
    seg = rows[a + 1:b + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("16 steps: %d kernels, span %.3f ms (%.3f ms per step), inside kernels %.3f ms, gaps %.3f ms (%.2f us per kernel)" %
          (len(seg), (t1 - t0) / 1e6, (t1 - t0) / 16e6, busy / 1e6, (t1 - t0 - busy) / 1e6, (t1 - t0 - busy) / 1e3 / len(seg)))
    import collections
    d = collections.defaultdict(list)
    for r in seg:
        d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("  %-60s x%-5d %7.2f us" % (k, len(v) // 16, sum(v) / len(v) / 1e3))
