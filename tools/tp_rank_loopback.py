#!/usr/bin/env python3
"""One rank of a tensor-parallel group ALONE on the GPU, its exchange looped back to itself (fl_tune "debug_tp_loopback").

    python tools/tp_rank_loopback.py [--model mistral-7b] [--tp 8] [--prompt 512] [--steps 128] [--fused 1|0] [--profile]

The rank runs its real shard shapes, its real launches (graph-replayed decode step) and its real exchange code -- the all-reduce in
the o_proj / down_proj GEMV epilogues (comm_ll.h) or the one-shot kernels (--fused 0), the logits gather -- but every inbox entry is
its own, so nothing waits for a peer and no link latency is included: this is what ONE rank's step costs before a byte crosses
xGMI ("time ONE rank with the exchange looped back to itself").  The sums are sums of tp copies of the rank's own
partials: tokens are meaningless, only the time is read.  The switch exists in the EXPERIMENTAL build only:
    FL_LIB_PATH=fastllm_amd/lib/libfastllm_mi355x_exp.so python tools/tp_rank_loopback.py ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--tp", default="8")
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--batch", default="", help="also time fl_batch_decode with these many streams on the rank, e.g. 8,32")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ["FL_TP_FUSED_AR"] = "1" if args.fused else "0"
    os.environ.setdefault("FL_AR_TIMEOUT_MS", "3000")
    import torch
    import bench
    import fastllm_amd as fa
    from fastllm_amd import binding
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS[args.model]
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
    T, K = args.prompt, args.steps
    prompt = np.random.RandomState(1234).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    rows = []
    for tp in [int(x) for x in args.tp.split(",")]:
        fa.tune("debug_tp_loopback", 0 if tp == 1 else 1)
        kw = {} if tp == 1 else dict(tp_mode=binding.TP_MULTI_PROCESS, tp_size=tp, tp_rank=0)
        m = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16", device_ids=[0], **kw)
        c = m.new_cache(T + 2 * K + 80)
        first = m.forward_argmax(c, prompt, 0) % cfg["vocab_size"]
        toks = m.decode_greedy(c, first, T, 8)                       # warm-up + graph capture
        m.synchronize()
        t0 = time.perf_counter()
        toks = m.decode_greedy(c, int(toks[-1]), T + 8, K)
        m.synchronize()
        dt = (time.perf_counter() - t0) / K
        info = m.info()
        row = {"model": args.model, "tp": tp, "kv_len": "%d..%d" % (T + 8, T + 8 + K), "ms_per_step_one_rank": round(dt * 1e3, 4),
               "fused_all_reduce": int(info.fused_all_reduce), "ceiling_tokens_per_sec_before_links": round(1.0 / dt, 1)}
        print("%s tp=%d: %.4f ms per step for one rank (loopback, %s)" % (args.model, tp, dt * 1e3, "all-reduce in the GEMV epilogues" if info.fused_all_reduce else "one-shot kernels" if tp > 1 else "single GPU"), flush=True)
        if args.profile:
            m.profile_begin()
            m.decode_greedy(c, int(toks[-1]), T + 8 + K, 8)
            st = m.profile_end()
            row["kernel_us_per_step"] = round(sum(s["total_ms"] for s in st) * 1e3 / 8, 1)
            row["kernels"] = []
            for s in st:
                row["kernels"].append({"name": s["name"], "launches_per_step": s["launches"] / 8, "us_per_launch": round(s["total_ms"] * 1e3 / s["launches"], 2),
                                       "GBps": round(s["bytes"] / s["total_ms"] / 1e6, 1) if s["total_ms"] else None})
                print("    %-36s x%-4d %8.2f us/launch %8.1f GB/s" % (s["name"], s["launches"] // 8, s["total_ms"] * 1e3 / s["launches"],
                                                                        s["bytes"] / (s["total_ms"] * 1e-3) / 1e9 if s["total_ms"] else 0), flush=True)
        for B in [int(x) for x in args.batch.split(",") if x]:
            if tp == 1 and False:
                continue
            caches, firsts = [], []
            for i in range(B):
                ci = m.new_cache(T + 120)
                firsts.append(m.forward_argmax(ci, prompt, 0) % cfg["vocab_size"])
                caches.append(ci)
            bt = fa.Batch(m, caches)
            g = bt.decode(firsts, [T] * B, 8)
            m.synchronize()
            t0 = time.perf_counter()
            g = bt.decode([int(x[-1]) % cfg["vocab_size"] for x in g], [T + 8] * B, 32)
            m.synchronize()
            db = (time.perf_counter() - t0) / 32
            row["batch_%d_ms_per_step_one_rank" % B] = round(db * 1e3, 4)
            print("%s tp=%d: %d streams %.4f ms per step for one rank -> %.0f tokens/s ceiling before links" % (args.model, tp, B, db * 1e3, B / db), flush=True)
            if args.profile:
                m.profile_begin()
                bt.decode([int(x[-1]) % cfg["vocab_size"] for x in g], [T + 40] * B, 4)
                for s_ in m.profile_end():
                    print("    %-36s x%-4d %8.2f us/launch" % (s_["name"], s_["launches"] // 4, s_["total_ms"] * 1e3 / s_["launches"]), flush=True)
            bt.close()
            for ci in caches:
                ci.close()
        rows.append(row)
        c.close(); m.close()
    fa.tune("debug_tp_loopback", 0)
    if args.out:
        commit = open(os.path.join(ROOT, ".fl_commit")).read().strip() if os.path.exists(os.path.join(ROOT, ".fl_commit")) else None
        with open(args.out, "w") as f:
            json.dump({"what": "one rank's decode step with its exchange looped back to itself (tools/tp_rank_loopback.py); no link crossed",
                       "commit": commit, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
