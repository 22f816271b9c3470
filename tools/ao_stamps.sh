#!/bin/bash
# Run on the GPU box: where the time goes inside the fused decode attention + o_proj launch (FL_FUSE_OPROJ=1).
# Every workgroup records s_memrealtime at its section boundaries (FL_AO_STAMPS, eager launches only); launches 201-203
# of the process are summarised on stderr.  Usage: tools/ao_stamps.sh [decode steps (sets the cache size)] [FL_AO_WAVES]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
[ -n "$2" ] && export FL_AO_WAVES=$2
FL_FUSE_OPROJ=1 FL_GRAPH=0 FL_AO_STAMPS=1 FL_BENCH_BATCH=0 timeout -k 10 300 python3 bench.py --steps ${1:-256} --warmup 2 --no-cpu-baseline 2>&1 | grep -a -A9 "^attn_oproj" | tail -10
