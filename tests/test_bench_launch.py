"""bench.py as the driver invokes it: `python bench.py --gpus N` with NO launcher must start its ranks itself
(child process, never exec), relay rank 0's JSON line and propagate failures."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_self_launch_propagates_rank_failure_without_gpu():
    """No GPU here: both ranks exit with "bench.py needs a GPU"; the parent must come back non-zero, print no JSON line,
    and must not have needed torch or a launcher itself."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    p, js = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {}, timeout=300)
    assert p.returncode != 0
    assert js is None
    assert "self-launch" in p.stderr and "needs a GPU" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("start_level", [0, 1])
def test_two_ranks_on_one_gpu_no_launcher(start_level):
    """FL_BENCH_SAME_DEVICE=1 python bench.py --gpus 2: the 2-rank path end to end on the box's one GPU (IPC-mapped
    inboxes; RCCL refuses two ranks per device, so fallback level 2 = "RCCL only" cannot run in this rehearsal: its
    plumbing is covered by test_rccl_plumbing_single_rank)."""
    p, js = run_bench(["--gpus", "2", "--steps", "8", "--warmup", "2", "--model", "tinyllama-1.1b", "--prompt", "64",
                       "--no-cpu-baseline"],
                      {"FL_BENCH_SAME_DEVICE": "1", "FL_BENCH_START_LEVEL": str(start_level), "FL_BENCH_BATCH": "0"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert js["n_gpus"] == 2 and js["ranks_agree"] is True
    assert js["config"]["tp_fallback_level"] == start_level
    assert js["config"]["rccl_ranks"] == 0                     # IPC-only group
    assert js["parity_check"]["ok"] and js["parity_check"]["tp_vs_single_gpu_rel_l2"] <= 5e-2
    assert js["value"] > 0 and js["steps"] == 8


@pytest.mark.gpu
def test_failed_level0_falls_back_to_level1():
    """One rank's health check fails at level 0 (injected: a real failure there is a bounded wait that gave up or an RCCL
    error): EVERY rank must rebuild one level more conservative and the run must complete at level 1."""
    p, js = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--model", "tinyllama-1.1b", "--prompt", "64",
                       "--no-cpu-baseline"],
                      {"FL_BENCH_SAME_DEVICE": "1", "FL_BENCH_INJECT_LEVEL0_FAILURE": "1", "FL_BENCH_BATCH": "0"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert "health check failed at level 0" in p.stderr
    assert js["n_gpus"] == 2 and js["ranks_agree"] is True and js["config"]["tp_fallback_level"] == 1
