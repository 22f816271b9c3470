"""Long tensor-parallel decode on ranks that share this GPU: the all-reduce fused into the GEMV epilogues must give the
same greedy and sampled tokens as the kernel form over hundreds of steps (epoch / half reuse), on every rank."""
import os, sys, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_tp_ipc as t
for name, tp, dtype in (("llama_tp4", 4, "bf16"), ("llama_tp4", 2, "bf16"), ("qwen2_a", 2, "f32")):
    out = {}
    for tag, fused in (("fused", "2"), ("plain", "0")):
        d = pathlib.Path(tempfile.mkdtemp())
        out[tag] = t.run_group(d, name, dtype, tp, 12, 2, 380, env_extra={"TP_WORKER_SAMPLED": "380", "FL_TP_FUSED_AR": fused})
    ok = all(np.array_equal(out["fused"][r][k], out["plain"][0][k]) for r in range(tp) for k in ("tokens", "sampled"))
    print(name, tp, dtype, "380 greedy + 380 sampled steps, fused == kernel form on all ranks:", ok, flush=True)
