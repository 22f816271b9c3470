#!/usr/bin/env python3
"""Probe: which factor breaks the fused (LL) all-reduce at 8 ranks as 4 processes x 2 rank threads?  Runs the worker groups of
tests/test_gpu_tp_ipc.py in a few variants and prints pass / the first error of each."""
import os
import sys
import tempfile
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_tp_ipc as t   # noqa: E402

variants = [
    ("tp4 rpp1 fused", "llama_tp4", 4, 1, {"FL_TP_FUSED_AR": "2", "TP_WORKER_TUNE": "gemv_blocks=16,gemv_waves=4"}),
    ("tp4 rpp2 fused", "llama_tp4", 4, 2, {"FL_TP_FUSED_AR": "2", "TP_WORKER_TUNE": "gemv_blocks=16,gemv_waves=4"}),
    ("tp2 rpp2 fused", "llama_tp4", 2, 2, {"FL_TP_FUSED_AR": "2", "TP_WORKER_TUNE": "gemv_blocks=16,gemv_waves=4"}),
    ("tp8 rpp2 fused nograph", "llama_tp8", 8, 2, {"FL_TP_FUSED_AR": "2", "FL_TP_GRAPH": "0", "TP_WORKER_TUNE": "gemv_blocks=16,gemv_waves=4"}),
    ("tp8 rpp2 fused blocks4", "llama_tp8", 8, 2, {"FL_TP_FUSED_AR": "2", "TP_WORKER_TUNE": "gemv_blocks=4,gemv_waves=4"}),
    ("tp8 rpp2 oneshot", "llama_tp8", 8, 2, {"FL_TP_FUSED_AR": "0", "TP_WORKER_TUNE": "gemv_blocks=16,gemv_waves=4"}),
]
only = sys.argv[1:] 
for name, model, tp, rpp, env in variants:
    if only and not any(o in name for o in only):
        continue
    d = pathlib.Path(tempfile.mkdtemp(prefix="tp8probe_", dir="/tmp"))
    try:
        t.run_group(d, model, "bf16", tp, 12, 2, 24, env_extra=dict(env, FL_AR_TIMEOUT_MS="3000"), ranks_per_proc=rpp)
        print("PASS", name, flush=True)
    except AssertionError as e:
        msg = str(e)
        errs = [ln for ln in msg.splitlines() if "error" in ln.lower() or "0xa11" in ln]
        print("FAIL", name, "|", errs[-1][-200:] if errs else msg[-300:], flush=True)
