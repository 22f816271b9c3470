#!/bin/bash
# Final pass of a round: the three bench lines and the rocprofv3 kernel stats from the last commit (the sweeps and PMC
# passes of tools/collect_profiles.sh do not change with host-side edits).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final3
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_mistral7b.json 2> $O/bench_mistral7b.err || exit 1
timeout -k 10 300 python3 bench.py --model tinyllama-1.1b --prompt 128 --steps 128 > $O/bench_tinyllama.json 2> $O/bench_tinyllama.err || exit 1
timeout -k 10 400 python3 bench.py --model qwen2-7b --prompt 4096 --steps 64 --no-cpu-baseline > $O/bench_qwen2_7b_4k.json 2> $O/bench_qwen2.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 64 --no-cpu-baseline > $O/prof_stats_bench.json 2> $O/prof_stats.err || exit 1
find $O -name '*kernel_trace.csv' -delete
ls -la $O
