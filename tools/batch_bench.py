"""Aggregate decode throughput of fl_batch (B concurrent streams) vs B = 1, synthetic weights in HBM."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--batches", default="1,2,4,8")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--dtype", default="bf16", help="bf16 | f32 (fp32 batches: RoPE / attention per sequence, the projections once for all rows)")
    args = ap.parse_args()
    import torch
    import bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS[args.model]
    dev = torch.device("cuda", 0)
    wts = bench.synth_device_weights(torch, cfg, dev)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype=args.dtype)
    del wts
    torch.cuda.empty_cache()
    T, K = args.prompt, args.steps
    rs = np.random.RandomState(1234)
    for B in [int(x) for x in args.batches.split(",")]:
        caches, firsts = [], []
        for i in range(B):
            p = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
            c = gm.new_cache(T + 2 * K + 80)
            firsts.append(gm.forward_argmax(c, p, 0))
            caches.append(c)
        batch = fa.Batch(gm, caches)
        got = batch.decode(firsts, [T] * B, 8)                       # warm-up + graph capture
        firsts = [int(g[-1]) for g in got]
        gm.synchronize()
        t0 = time.perf_counter()
        got = batch.decode(firsts, [T + 8] * B, K)
        gm.synchronize()
        dt = time.perf_counter() - t0
        print("B=%d: %.3f ms/step, %.1f tokens/s aggregate, %.1f per stream" % (B, dt / K * 1e3, B * K / dt, K / dt), flush=True)
        if args.profile:
            gm.profile_begin()
            batch.decode([int(g[-1]) for g in got], [T + 8 + K] * B, 4)
            for s in gm.profile_end():
                print("    %-34s x%-4d %9.2f us/launch %8.1f GB/s" % (s["name"], s["launches"] // 4, s["total_ms"] / s["launches"] * 1e3,
                                                                       s["bytes"] / (s["total_ms"] * 1e-3) / 1e9 if s["total_ms"] else 0))
        batch.close()
        for c in caches:
            c.close()


if __name__ == "__main__":
    main()
