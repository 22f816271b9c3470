#!/usr/bin/env python3
"""Long greedy runs through the persistent decode engine (FL_ENGINE=1) and through the short-cache replicated attention: the same
ids twice from the same state (no atomics, fixed summation orders), no device-side wait giving up, over thousands of graph replays.

    python tools/engine_soak.py [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    import torch
    import bench
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["tinyllama-1.1b"]
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
    prompt = np.random.RandomState(5).randint(0, cfg["vocab_size"], size=32).astype(np.uint32)
    for env, cap in (({"FL_ENGINE": "1"}, 32 + steps + 8), ({"FL_ENGINE": "0"}, 96)):
        os.environ.update(env)
        m = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
        n = min(steps, cap - 40)
        runs = []
        t0 = time.perf_counter()
        for _ in range(2):
            c = m.new_cache(cap)
            first = m.forward_argmax(c, prompt, 0)
            runs.append(m.decode_greedy(c, first, len(prompt), n))
            c.close()
        dt = time.perf_counter() - t0
        ok = bool((runs[0] == runs[1]).all())
        print("%s cache %d: 2 x %d greedy steps in %.1f s, identical ids: %s, distinct tokens %d" % (env, cap, n, dt, ok, len(set(runs[0].tolist()))), flush=True)
        assert ok
        m.close()


if __name__ == "__main__":
    main()
