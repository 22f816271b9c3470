"""Wall time of one prefill (no per-launch events) beside the sum of its kernels: run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prefill_wall.py MODEL T`, then
`python3 tools/prefill_wall.py --trace DIR` prints the last call's span, busy time and largest gaps."""
import os, sys, glob, csv, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--trace":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    # the last prefill = the launches after the last gap longer than 2 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 2_000_000:
            cut = i
    rows = rows[cut:]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = sorted(((rows[i][0] - rows[i - 1][1], rows[i - 1][2][:50], rows[i][2][:50]) for i in range(1, len(rows))), reverse=True)
    print("launches %d  span %.3f ms  kernels %.3f ms  gaps %.3f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for g in gaps[:8]:
        print("  gap %.1f us  after %s  before %s" % (g[0] / 1e3, g[1], g[2]))
    byk = {}
    for s, e, k in rows:
        k = k[:70]
        byk.setdefault(k, [0, 0]); byk[k][0] += 1; byk[k][1] += e - s
    for k, (n, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-70s x%-4d %8.3f ms" % (k, n, t / 1e6))
    sys.exit(0)
import numpy as np
import torch, bench
import fastllm_amd as fa
from fastllm_amd.configs import MODEL_CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = dict(MODEL_CONFIGS[name])
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
p = np.random.RandomState(0).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
c = gm.new_cache(T + 8)
for i in range(4):
    c.reset(); gm.synchronize(); time.sleep(0.01)
    t0 = time.perf_counter(); gm.forward_argmax(c, p, 0); gm.synchronize()
    print("%s T=%d call %d: %.3f ms wall" % (name, T, i, (time.perf_counter() - t0) * 1e3), flush=True)
