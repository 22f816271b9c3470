#!/bin/bash
run() { FL_BENCH_BATCH=0 timeout -k 10 300 python3 bench.py --model $1 --prompt $2 --steps $3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$4', '$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_check']['rel_l2'], [ (k['name'].replace('gemv',''), k['us_per_launch']) for k in d['kernels'] if 'gemv' in k['name']])
"; }
for rep in 1 2; do
FL_LIB_PATH=$PWD/tools/_old.so run $1 $2 $3 old
run $1 $2 $3 new
done
