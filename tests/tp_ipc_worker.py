"""One rank -- or several, "0,1", each on its own thread -- of a multi-process tensor-parallel group
(tests/test_gpu_tp_ipc.py starts these).  Several ranks per process is how an 8-rank group runs on a one-GPU box that admits
at most six GPU processes: 4 processes x 2 ranks; the peers of another process are reached through real hipIpc mappings, the
one in the same process through its plain pointer (comm.hip, comm_exported_here).

No RCCL: the ranks may share one GPU (the GPU boxes have one), so the group is wired with
fl_comm_ipc_export / fl_comm_ipc_connect and every collective takes the one-shot path over
peer-mapped inboxes.  Handles are exchanged through files in `outdir`.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def exchange(outdir, rank, world, blob, tag, timeout=120.0):
    tmp = os.path.join(outdir, "%s_%d.tmp" % (tag, rank))
    with open(tmp, "wb") as f:
        f.write(blob)
    os.rename(tmp, os.path.join(outdir, "%s_%d" % (tag, rank)))
    out, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(outdir, "%s_%d" % (tag, r))
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d never published %s" % (r, tag))
            time.sleep(0.01)
        with open(p, "rb") as f:
            out.append(f.read())
    return out


def main():
    ranks, world, name, dtype, outdir = [int(r) for r in sys.argv[1].split(",")], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    T, n_tf, n_greedy = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    if len(ranks) == 1:
        return run_rank(ranks[0], world, name, dtype, outdir, T, n_tf, n_greedy)
    import threading
    import traceback
    failed = []

    def guarded(r):
        try:
            run_rank(r, world, name, dtype, outdir, T, n_tf, n_greedy)
        except BaseException:                                # noqa: reported below, the process exits non-zero
            traceback.print_exc()
            failed.append(r)
    ts = [threading.Thread(target=guarded, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if failed:
        raise SystemExit("ranks %s failed" % failed)


def run_rank(rank, world, name, dtype, outdir, T, n_tf, n_greedy):
    import fastllm_amd as fa
    from fastllm_amd import binding
    import synth
    cfg = synth.CONFIGS[name] if name in synth.CONFIGS else fa.MODEL_CONFIGS[name]
    for kv in filter(None, os.environ.get("TP_WORKER_TUNE", "").split(",")):       # e.g. gemv_blocks=8,gemv_waves=4
        k, v = kv.split("=")
        fa.tune(k, int(v))
    w = synth.synth_weights(cfg)
    m = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_MULTI_PROCESS, tp_size=world, tp_rank=rank, device_ids=[0])
    m.ipc_connect(exchange(outdir, rank, world, m.ipc_export(), "handle"))
    ids = synth.prompt_ids(cfg, T + n_tf, seed=11)
    c = m.new_cache(T + n_tf + n_greedy + 8)
    selftest = [m.comm_selftest(n) for n in (4096, 16384, 98304)] if os.environ.get("TP_WORKER_SELFTEST", "") == "1" else []
    res = {"prefill": m.forward(c, ids[:T], 0)}
    if selftest:
        res["selftest"] = np.array(selftest)
    if os.environ.get("TP_WORKER_PROFILE", "") == "1":
        # the same prompt again under the profiler: launches bracketed by events, which also keeps the all-reduces on the compute
        # stream -- the schedule FL_TP_EMULATED runs, so these logits are the emulated group's bits
        c2 = m.new_cache(T + 8)
        m.profile_begin()
        res["prefill_profiled"] = m.forward(c2, ids[:T], 0)
        res["prefill_kernels"] = np.array(sorted(set(s["name"] for s in m.profile_end())))
        c2.close()
    stalled = os.environ.get("TP_WORKER_STALLED_RANK", "")
    if stalled:
        # rank `stalled` never issues the decode step (but stays mapped: nobody may push into freed memory); the others
        # must come back with an error after the bounded wait instead of hanging the GPU
        n_batch = int(os.environ.get("TP_WORKER_BATCH", "0"))
        bt = None
        if n_batch:                                      # the step that stalls is a BATCH step: [B, h] all-reduces on the many-workgroup collective
            caches, firsts = [], []
            for i in range(n_batch):
                ci = m.new_cache(32)
                firsts.append(m.forward_argmax(ci, synth.prompt_ids(cfg, 4 + i, seed=70 + i), 0))
                caches.append(ci)
            bt = fa.Batch(m, caches)
        if rank != int(stalled):
            t0 = time.time()
            try:
                if bt is not None:
                    bt.forward(firsts, [4 + i for i in range(n_batch)])
                else:
                    m.decode_greedy(c, 1, T, 1)
                msg = "no error"
            except RuntimeError as e:
                msg = str(e)
            res["error"] = np.array(msg)
            res["waited_s"] = np.array(time.time() - t0)
        np.savez(os.path.join(outdir, "out_%d.npz" % rank), **res)
        exchange(outdir, rank, world, b"done", "done")
        return
    step = []
    for i in range(T, T + n_tf):
        step.append(m.forward(c, ids[i:i + 1], i))
    res["decode"] = np.stack(step) if step else np.zeros((0, cfg["vocab_size"]), np.float32)
    first = m.forward_argmax(c, ids[:1], T + n_tf)
    t0 = time.time()
    rest = m.decode_greedy(c, first, T + n_tf + 1, n_greedy)
    res["greedy_s"] = np.array(time.time() - t0)
    res["tokens"] = np.concatenate([[first], rest]).astype(np.uint32)
    n_sampled = int(os.environ.get("TP_WORKER_SAMPLED", "0"))
    if n_sampled:                                        # temperature sampling follows the logits bit for bit
        c2 = m.new_cache(T + n_sampled + 8)
        f2 = m.forward_sample(c2, ids[:T], 0, 0.9, seed=5)
        res["sampled"] = np.concatenate([[f2], m.decode_sample(c2, f2, T, n_sampled, 0.9, seed=5)]).astype(np.uint32)
        c2.close()
    n_batch = int(os.environ.get("TP_WORKER_BATCH", "0"))
    if n_batch:                                          # a batch of streams on the group: every rank builds the same batch
        lens = [3 + (2 * i) % 11 for i in range(n_batch)]
        n_bsteps = int(os.environ.get("TP_WORKER_BATCH_STEPS", "10"))
        caches, firsts = [], []
        for i, n in enumerate(lens):
            ci = m.new_cache(38 + n_bsteps)
            firsts.append(m.forward_argmax(ci, synth.prompt_ids(cfg, n, seed=70 + i), 0))
            caches.append(ci)
        bt = fa.Batch(m, caches)
        lg, am = bt.forward(firsts, lens)
        res["batch_logits"], res["batch_first"] = lg, np.asarray(firsts, dtype=np.uint32)
        toks = bt.decode([int(t) for t in am], [n + 1 for n in lens], n_bsteps)
        res["batch_tokens"] = np.stack([np.concatenate([[am[i]], toks[i]]).astype(np.uint32) for i in range(n_batch)])
        if os.environ.get("TP_WORKER_BATCH_REPLACE", "") == "1":      # continuous batching on the group: every rank swaps the same slot
            cn = m.new_cache(38 + n_bsteps)
            fn = m.forward_argmax(cn, synth.prompt_ids(cfg, 9, seed=333), 0)
            bt.replace(1, cn)
            first2 = [int(toks[i][-1]) for i in range(n_batch)]
            pos2 = [lens[i] + 1 + n_bsteps for i in range(n_batch)]
            first2[1], pos2[1] = fn, 9
            t2 = bt.decode(first2, pos2, 6)
            res["batch_tokens_after_replace"] = np.stack([np.asarray(t2[i], dtype=np.uint32) for i in range(n_batch)])
            res["batch_replaced_first"] = np.array(fn, dtype=np.uint32)
            caches.append(cn)
        bt.close()
        for ci in caches:
            ci.close()
    np.savez(os.path.join(outdir, "out_%d.npz" % rank), **res)
    exchange(outdir, rank, world, b"done", "done")      # nobody unmaps an inbox a peer may still push to
    c.close()
    m.close()


if __name__ == "__main__":
    main()
