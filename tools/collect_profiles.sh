#!/bin/bash
# Run on the GPU box (via gpurun): bench lines + rocprofv3 kernel stats + PMC passes into gpurun_out/final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_mistral7b.json 2> $O/bench_mistral7b.err || exit 1
timeout -k 10 300 python3 bench.py --model tinyllama-1.1b --prompt 128 --steps 128 > $O/bench_tinyllama.json 2> $O/bench_tinyllama.err || exit 1
timeout -k 10 400 python3 bench.py --model qwen2-7b --prompt 4096 --steps 64 --no-cpu-baseline > $O/bench_qwen2_7b_4k.json 2> $O/bench_qwen2.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 64 --no-cpu-baseline > $O/prof_stats_bench.json 2> $O/prof_stats.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 16 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 16 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || exit 1
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_gemv.json > /dev/null
# keep the merged output small: the per-dispatch traces are large
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O
