#!/bin/bash
run() { FL_BENCH_BATCH=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], [ (k['name'].replace('gemv',''), k['us_per_launch']) for k in d['kernels'] if 'gemv' in k['name']])
"; }
run default
FL_GEMV_U=8 run U=8
FL_GEMV_U=2 run U=2
FL_GEMV_R=4 FL_GEMV_U=2 run R=4,U=2
run default
