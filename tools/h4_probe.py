#!/usr/bin/env python3
"""The 128 x 256 GEMM with in-launch K-slice sums (k_gemm_h4.hip): correctness on integer operands, repeatability, and time
per launch beside the kernels the selection uses today (GPU box).  usage: h4_probe.py [check|time|all]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fastllm_amd as fa
import synth

what = sys.argv[1] if len(sys.argv) > 1 else "all"


def ints(shape, seed):
    return np.random.RandomState(seed).randint(-3, 4, size=shape).astype(np.float32)


if what in ("check", "all"):
    bad = 0
    for (T, N, K) in [(128, 256, 64), (128, 256, 128), (128, 256, 192), (128, 256, 448), (130, 300, 512), (512, 4096, 4096), (512, 6144, 4096),
                      (512, 4096, 14336), (257, 1000, 1024), (1000, 4096, 2048), (300, 520, 6400)]:
        x, w = ints((T, K), T + K), ints((N, K), N + K)
        ref = (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
        xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
        for ks in (1, 2, 3, 4, -2, -3, -4):                 # negative: early slices abandon their blocks at once (the rescue path)
            fa.tune("h4_wait_us", 0 if ks < 0 else 30)
            ks = abs(ks)
            if K // 64 < ks:
                continue
            fa.tune("gemm_h4", 2); fa.tune("h4_split", ks)
            y = fa.op_linear(xb, wb, None)
            ok = np.array_equal(y, ref)
            rep = all(np.array_equal(fa.op_linear(xb, wb, None), y) for _ in range(5))
            nbad = int((y != ref).sum())
            print("T=%4d N=%5d K=%5d slices=%d  %s  repeat %s  (%d wrong)" % (T, N, K, ks, "exact" if ok else "WRONG", "same" if rep else "DIFFERS", nbad), flush=True)
            bad += (not ok) or (not rep)
    # bias + float data, gate/up
    rs = np.random.RandomState(5)
    for (T, N, K, epi) in [(512, 4096, 4096, 0), (200, 1408, 1024, 1), (384, 704, 512, 1)]:
        x = (rs.standard_normal((T, K))).astype(np.float32)
        w = (rs.standard_normal((N if not epi else 2 * N, K)) * 0.05).astype(np.float32)
        b = rs.standard_normal((N,)).astype(np.float32) if not epi else None
        xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
        xf, wf = synth.bf16_bits_to_f32(xb).astype(np.float64), synth.bf16_bits_to_f32(wb).astype(np.float64)
        ref = xf @ wf.T
        if b is not None:
            ref = ref + b
        if epi:
            g, u = ref[:, :N], ref[:, N:]
            ref = g / (1.0 + np.exp(-g)) * u
        for ks in (1, 2, 4):
            fa.tune("gemm_h4", 2); fa.tune("h4_split", ks)
            y = fa.op_linear(xb, wb, b, epilogue=epi)
            err = (np.abs(y - ref) / (1.0 + np.abs(ref) * (2.0 if epi else 0.0))).max()   # gate/up: the output is rounded to bf16
            tol = 2.0 ** -8 if epi else 2e-5 * np.sqrt(K) + 1e-5
            print("float T=%d N=%d K=%d epi=%d slices=%d  max err %.3g (tol %.3g) %s" % (T, N, K, epi, ks, err, tol, "ok" if err <= tol else "WRONG"), flush=True)
            bad += err > tol
    print("CHECK", "FAILED" if bad else "passed", flush=True)
    if bad:
        sys.exit(1)

fa.tune("h4_wait_us", 30)
if what in ("time", "all"):
    SHAPES = [("mistral qkv", 512, 6144, 4096, 0), ("mistral o", 512, 4096, 4096, 0), ("mistral down", 512, 4096, 14336, 0),
              ("mistral gate/up", 512, 28672, 4096, 1), ("mistral o 256", 256, 4096, 4096, 0), ("mistral down 256", 256, 4096, 14336, 0),
              ("mistral o 1024", 1024, 4096, 4096, 0), ("mistral down 1024", 1024, 4096, 14336, 0), ("mistral qkv 1024", 1024, 6144, 4096, 0),
              ("qwen2 o 512", 512, 3584, 3584, 0), ("qwen2 down 512", 512, 3584, 18944, 0), ("qwen2 qkv 512", 512, 4608, 3584, 0),
              ("tinyllama o 512", 512, 2048, 2048, 0), ("tinyllama down 512", 512, 2048, 5632, 0)]
    rs = np.random.RandomState(0)
    for name, T, N, K, epi in SHAPES:
        w = ((rs.randint(0, 65536, size=(N, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        x = ((rs.randint(0, 65536, size=(T, K), dtype=np.uint16)) & 0x807F) | 0x3C00
        fa.tune("gemm_h4", 0)
        _, ms0 = fa.op_linear(x, w, None, epilogue=epi, iters=20)
        line = "%-20s T=%4d N=%5d K=%5d  today %7.1f us (+ slab sum) |" % (name, T, N, K, ms0 * 1e3)
        for ks in (1, 2, 3, 4):
            fa.tune("gemm_h4", 2); fa.tune("h4_split", ks)
            _, ms = fa.op_linear(x, w, None, epilogue=epi, iters=20)
            line += " h4/%d %7.1f us %6.0f TF |" % (ks, ms * 1e3, 2.0 * T * N * K / ms / 1e9)
        print(line, flush=True)
