#!/usr/bin/env python3
"""One model's decode step under the environment it is started with: tokens/s over K greedy steps (hipGraph replay) and the
per-kernel HIP-event times of 8 eager steps.   python tools/decode_probe.py [model] [prompt] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    import torch
    import bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name])
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
    prompt = np.random.RandomState(1234).randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    prompt[0] = 1
    m = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    c = m.new_cache(T + 2 * K + 80)
    first = m.forward_argmax(c, prompt, 0)
    toks = m.decode_greedy(c, first, T, 16)
    m.synchronize()
    best = 0.0
    pos = T + 16
    for _ in range(2):
        t0 = time.perf_counter()
        toks = m.decode_greedy(c, int(toks[-1]), pos, K)
        m.synchronize()
        best = max(best, K / (time.perf_counter() - t0))
        pos += K
        if pos + K + 16 > T + 2 * K + 64:
            break
    print("%s prompt %d: %.1f tokens/s (%.1f us/step)  ids tail %s" % (name, T, best, 1e6 / best, toks[-4:]))
    m.profile_begin()
    m.decode_greedy(c, int(toks[-1]), pos, 8)
    for s in m.profile_end():
        print("   %-40s x%-4d %9.2f us/step %8.2f us/launch %8.1f GB/s" % (
            s["name"], s["launches"] // 8, s["total_ms"] * 1e3 / 8, s["total_ms"] * 1e3 / s["launches"],
            s["bytes"] / s["total_ms"] / 1e6 if s["total_ms"] else 0))


if __name__ == "__main__":
    main()
