"""Long batched decode on the ranks of a multi-process group that share this GPU: hundreds of steps of [B, h] all-reduces on the
many-workgroup one-shot collective (epoch / inbox-half reuse, slice flags, the last workgroup's epoch ticket) must leave every rank with
the same tokens, twice over (the kernels are deterministic).  usage: tp_batch_soak.py [steps]"""
import os, sys, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_tp_ipc as t
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, tp, B in (("mistral_wide", 4, 16), ("mistral_wide", 2, 32), ("llama_tp8", 8, 16)):
    runs = []
    for rep in range(2):
        d = pathlib.Path(tempfile.mkdtemp())
        runs.append(t.run_group(d, name, "bf16", tp, env_extra={"TP_WORKER_BATCH": str(B), "TP_WORKER_BATCH_STEPS": str(steps), "FL_ATTN_REP": "0" if tp > 4 else "1"}))
    ok = all(np.array_equal(runs[rep][r]["batch_tokens"], runs[0][0]["batch_tokens"]) for rep in range(2) for r in range(tp))
    n_coll = steps * 2 * 2 + steps
    print("%s tp=%d B=%d: %d batch steps (%d one-shot collectives per rank), tokens equal on all ranks and in both runs: %s" % (name, tp, B, steps, n_coll, ok), flush=True)
    assert ok
