#!/usr/bin/env python3
"""Run one prefill GEMM shape through fl_op_linear with the stamped 8p instantiation.  Usage: FL_8P_STAMPS=out python tools/stamps_run.py T N K epi"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastllm_amd as fa
T, N, K, epi = (int(v) for v in sys.argv[1:5])
rs = np.random.RandomState(0)
w = (rs.randint(0, 65536, size=(N, K), dtype=np.uint16) & 0x807F) | 0x3C00
x = (rs.randint(0, 65536, size=(T, K), dtype=np.uint16) & 0x807F) | 0x3C00
for _ in range(3):
    fa.op_linear(x, w, None, epilogue=epi)
