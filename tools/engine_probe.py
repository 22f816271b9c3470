#!/usr/bin/env python3
"""The persistent decode engine (k_engine.hip) against the launch-per-projection decode step on one model:
same weights, same prompt -- logits side by side, greedy ids, tokens/s, per-kernel HIP-event times.

    python tools/engine_probe.py [model] [prompt] [steps] [layers]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tinyllama-1.1b"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    import torch
    import bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name])
    if len(sys.argv) > 4:
        cfg["num_hidden_layers"] = int(sys.argv[4])
    dev = torch.device("cuda", 0)
    wts = bench.synth_device_weights(torch, cfg, dev)
    prompt = np.random.RandomState(1234).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    prompt[0] = 1
    res = {}
    for mode in ("0", "1"):
        os.environ["FL_ENGINE"] = mode
        fa.reload_env()
        m = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
        c = m.new_cache(T + K + 80)
        lg0 = m.forward(c, prompt, 0)
        first = int(np.argmax(lg0))
        steps = [m.forward(c, [first], T)]                               # one eager step with logits
        toks = m.decode_greedy(c, int(np.argmax(steps[0])), T + 1, 8)  # warm-up + graph capture
        m.synchronize()
        t0 = time.perf_counter()
        toks2 = m.decode_greedy(c, int(toks[-1]), T + 9, K)
        m.synchronize()
        dt = time.perf_counter() - t0
        m.profile_begin()
        m.decode_greedy(c, int(toks2[-1]), T + 9 + K, 8)
        st = m.profile_end()
        res[mode] = (steps[0], np.concatenate([toks, toks2]), K / dt, st)
        print("FL_ENGINE=%s: %.1f tokens/s (%.1f us/step)" % (mode, K / dt, dt / K * 1e6))
        for s in st:
            print("   %-40s x%-4d %9.2f us/step %8.2f us/launch %8.1f GB/s" % (
                s["name"], s["launches"] // 8, s["total_ms"] * 1e3 / 8, s["total_ms"] * 1e3 / s["launches"],
                s["bytes"] / s["total_ms"] / 1e6 if s["total_ms"] else 0))
        c.close(); m.close()
    a, b = res["0"][0], res["1"][0]
    print("decode-step logits, engine vs launches: rel L2 %.3e, max |diff| %.3e, argmax %d / %d" % (
        np.linalg.norm(a - b) / np.linalg.norm(a), np.abs(a - b).max(), int(np.argmax(b)), int(np.argmax(a))))
    ta, tb = res["0"][1], res["1"][1]
    same = int(np.argmin(ta == tb)) if not (ta == tb).all() else len(ta)
    print("greedy ids: %d / %d equal from the start" % (same, len(ta)))
    print("speed-up: %.3fx" % (res["1"][2] / res["0"][2]))


if __name__ == "__main__":
    main()
