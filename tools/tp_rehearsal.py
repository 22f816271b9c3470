#!/usr/bin/env python3
"""tools/tp_rehearsal.py -- an N-rank tensor-parallel group on ONE GPU with several ranks per process.

    python tools/tp_rehearsal.py --ranks 8 --procs 4 --model mistral-7b --prompt 512 --steps 64 [--fused] [--out FILE]

Why this exists beside `FL_BENCH_SAME_DEVICE=1 python bench.py --gpus N`: a GPU box of the pool admits at most SIX processes on
its card, so bench.py's one-process-per-rank launch cannot rehearse BASELINE config C4 (Mistral-7B, TP = 8) there.  Here the 8 ranks
are 4 processes x 2 rank threads: every rank is a separate FL_TP_MULTI_PROCESS model (what bench.py creates), reaches the six peers of
the other processes through real hipIpc mappings and the one in its own process through its plain pointer (comm.hip), and runs what
bench.py runs -- health check, parity against one GPU running the whole model, prefill, K greedy decode steps, per-rank kernel times --
plus one thing bench.py does not: the same prompt through FL_TP_EMULATED, whose tokens must be identical.
All ranks share one HBM: the tokens/s printed is NOT a scaling figure.  `--fused` rehearses the all-reduce in the GEMV epilogues
(comm_ll.h) with every rank's GEMV grid cut to 1/N of the card, as FL_BENCH_SAME_DEVICE_FUSED=1 does.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time
import traceback
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def publish(outdir, tag, rank, blob):
    tmp = os.path.join(outdir, "%s_%d.tmp" % (tag, rank))
    with open(tmp, "wb") as f:
        f.write(blob)
    os.rename(tmp, os.path.join(outdir, "%s_%d" % (tag, rank)))


def gather(outdir, tag, world, timeout=900.0):
    out, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(outdir, "%s_%d" % (tag, r))
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d never published %s" % (r, tag))
            time.sleep(0.002)
        with open(p, "rb") as f:
            out.append(f.read())
    return out


def exchange(outdir, tag, rank, world, blob=b"x"):
    publish(outdir, tag, rank, blob)
    return gather(outdir, tag, world)


def rank_main(args, rank, wts, shared, results):
    import torch
    import fastllm_amd as fa
    from fastllm_amd import binding
    import bench
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS[args.model]
    world, outdir, T, K = args.ranks, args.dir, args.prompt, args.steps
    bar = lambda tag: exchange(outdir, tag, rank, world)      # noqa: E731
    m = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", tp_mode=binding.TP_MULTI_PROCESS, tp_size=world, tp_rank=rank, device_ids=[0])
    m.ipc_connect(exchange(outdir, "handle", rank, world, m.ipc_export()))
    res = {"rank": rank}
    # health check, as bench.py: one prefill of the workload's length + four decode steps
    hp = np.random.RandomState(7).randint(0, cfg["vocab_size"], size=max(T, 8)).astype(np.uint32)
    hc = m.new_cache(len(hp) + 16)
    hf = m.forward_argmax(hc, hp, 0)
    m.decode_greedy(hc, hf, len(hp), 4)
    hc.close()
    bar("health")
    # the group against ONE GPU running the whole model (bench.py's tp_vs_single_gpu_rel_l2), and against the emulated group (same bits)
    hp = np.random.RandomState(11).randint(0, cfg["vocab_size"], size=32).astype(np.uint32)
    hc = m.new_cache(48)
    lt = m.forward(hc, hp, 0)
    hc.close()
    rs = np.random.RandomState(1234)
    prompt = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    prompt[0] = 1
    cache = m.new_cache(T + K + 72)
    first = m.forward_argmax(cache, prompt, 0)
    m.decode_greedy(cache, first, T, 8)                        # warm-up + graph capture
    samples = []
    for i in range(3):
        cache.reset()
        bar("pre%d" % i); m.synchronize()
        t0 = time.perf_counter()
        first = m.forward_argmax(cache, prompt, 0)
        m.synchronize()
        samples.append(time.perf_counter() - t0)
    res["prefill_ms"] = sorted(samples)[1] * 1e3
    bar("dec0"); m.synchronize()
    t0 = time.perf_counter()
    toks = m.decode_greedy(cache, first, T, K)
    m.synchronize()
    res["decode_s"] = time.perf_counter() - t0
    bar("dec1")
    res["tokens_crc32"] = zlib.crc32(np.concatenate([[first], toks]).astype(np.uint32).tobytes())
    res["first_tokens"] = [int(first)] + [int(t) for t in toks[:7]]
    n_prof = 4
    m.profile_begin()
    m.decode_greedy(cache, int(toks[-1]), T + K, n_prof)
    stats = m.profile_end()
    res["kernel_us_per_step"] = round(sum(s["total_ms"] for s in stats) * 1e3 / n_prof, 1)
    res["kernels"] = [{"name": s["name"], "launches_per_step": s["launches"] / n_prof, "us_per_launch": round(s["total_ms"] * 1e3 / s["launches"], 2)} for s in stats]
    cache.reset()
    m.profile_begin()
    m.forward_argmax(cache, prompt, 0)
    res["prefill_kernels"] = sorted(set(s["name"] for s in m.profile_end()))
    info = m.info()
    res["fused_all_reduce"], res["small_collectives"] = int(info.fused_all_reduce), int(info.small_collectives)
    bar("prof")
    if rank == 0:
        sm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", tp_mode=binding.TP_NONE, device_ids=[0])
        sc = sm.new_cache(48)
        ls = sm.forward(sc, hp, 0)
        sc.close(); sm.close()
        res["tp_vs_single_gpu_rel_l2"] = float(np.linalg.norm(lt - ls) / max(np.linalg.norm(ls), 1e-30))
        # the emulated group runs the same shards with the same summation order (and, for --fused, the same cut GEMV grids): same tokens
        em = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=world, device_ids=[0])
        ec = em.new_cache(T + K + 72)
        ef = em.forward_argmax(ec, prompt, 0)
        et = em.decode_greedy(ec, ef, T, K)
        res["emulated_crc32"] = zlib.crc32(np.concatenate([[ef], et]).astype(np.uint32).tobytes())
        ec.close(); em.close()
    bar("done")                                                # nobody unmaps an inbox a peer may still push to
    cache.close()
    m.close()
    results[rank] = res


def worker(args):
    import torch
    import fastllm_amd as fa
    import bench
    from fastllm_amd.configs import MODEL_CONFIGS
    assert torch.cuda.is_available()
    ranks = [int(r) for r in args.worker.split(",")]
    if args.fused:
        fa.tune("gemv_blocks", 192 // args.ranks)
        fa.tune("gemv_waves", 4)
    wts = bench.synth_device_weights(torch, MODEL_CONFIGS[args.model], torch.device("cuda", 0))
    results, failed = {}, []

    def guarded(r):
        try:
            rank_main(args, r, wts, None, results)
        except BaseException:                                  # noqa
            traceback.print_exc()
            failed.append(r)
    ts = [threading.Thread(target=guarded, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r, res in results.items():
        publish(args.dir, "result", r, json.dumps(res).encode())
    if failed:
        raise SystemExit("ranks %s failed" % failed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--worker", default=None)
    args = ap.parse_args()
    if args.worker is not None:
        return worker(args)
    import tempfile
    assert args.ranks % args.procs == 0 and args.procs <= 5, "at most five worker processes (the boxes admit six on the card)"
    rpp = args.ranks // args.procs
    args.dir = tempfile.mkdtemp(prefix="fl_tp_rehearsal_", dir="/tmp")
    env = dict(os.environ, FL_TP_OVERLAP=os.environ.get("FL_TP_OVERLAP", "0"), FL_TP_FUSED_AR="2" if args.fused else "0",
               FL_AR_TIMEOUT_MS=os.environ.get("FL_AR_TIMEOUT_MS", "20000"), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    procs = []
    for p in range(args.procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--ranks", str(args.ranks), "--procs", str(args.procs), "--model", args.model, "--prompt", str(args.prompt),
               "--steps", str(args.steps), "--dir", args.dir, "--worker", ",".join(str(r) for r in range(p * rpp, (p + 1) * rpp))] + (["--fused"] if args.fused else [])
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for p in procs:
        try:
            rc = p.wait(timeout=1100) or rc
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 124
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise SystemExit("rehearsal failed (exit code %d)" % rc)
    res = [json.loads(b) for b in gather(args.dir, "result", args.ranks, timeout=5)]
    crcs = [r["tokens_crc32"] for r in res]
    dec = max(r["decode_s"] for r in res)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    if not head and os.path.exists(os.path.join(ROOT, ".fl_commit")):        # the GPU box has no .git: tools/gpu.sh leaves the commit here
        head = open(os.path.join(ROOT, ".fl_commit")).read().strip()
    out = {"what": "tensor-parallel rehearsal: %d ranks as %d processes x %d rank threads on ONE GPU (real hipIpc inboxes between processes); shared HBM, NOT a scaling figure"
                   % (args.ranks, args.procs, rpp),
           "commit": head or None, "model": args.model, "prompt": args.prompt, "steps": args.steps, "n_ranks": args.ranks, "n_processes": args.procs,
           "fused_all_reduce": bool(res[0]["fused_all_reduce"]), "ranks_agree": all(c == crcs[0] for c in crcs),
           "tokens_equal_emulated": crcs[0] == res[0].get("emulated_crc32"), "tokens_crc32": crcs[0],
           "tp_vs_single_gpu_rel_l2": round(res[0]["tp_vs_single_gpu_rel_l2"], 6),
           "decode_all_reduce": "in the o_proj / down_proj GEMV epilogues (comm_ll.h), GEMV grids cut to 1/N of the card" if args.fused else "one-shot kernels (k_comm.hip)",
           "ms_per_step_shared_gpu": round(dec / args.steps * 1e3, 4), "prefill_ms_shared_gpu": round(max(r["prefill_ms"] for r in res), 2),
           "prefill_kernels": res[0]["prefill_kernels"],
           "per_rank": [{"rank": r["rank"], "kernel_us_per_step": r["kernel_us_per_step"], "kernels": r["kernels"]} for r in res]}
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    ok = out["ranks_agree"] and out["tokens_equal_emulated"] and out["tp_vs_single_gpu_rel_l2"] <= 5e-2
    raise SystemExit(0 if ok else 3)


if __name__ == "__main__":
    main()
