#!/bin/bash
# Round 3: run on the GPU box (via gpurun, two calls: part 1 | part 2): the bench lines, rocprofv3 kernel stats and the PMC passes
# of the FINAL commit into gpurun_out/final/ (copied to profiles/r03/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "$1" != "2" ]; then
timeout -k 10 400 python3 bench.py > $O/bench_mistral7b_decode.json 2> $O/bench_mistral7b.err || exit 1
timeout -k 10 300 python3 bench.py --model tinyllama-1.1b --prompt 128 --steps 128 > $O/bench_tinyllama_decode.json 2> $O/bench_tinyllama.err || exit 1
timeout -k 10 400 python3 bench.py --model qwen2-7b --prompt 4096 --steps 64 --no-cpu-baseline > $O/bench_qwen2_7b_4k.json 2> $O/bench_qwen2.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 64 --no-cpu-baseline > $O/prof_stats_bench.json 2> $O/prof_stats.err || exit 1
cd $R
python3 tools/rocprof_gemv.py $(find $O/prof_stats -name '*kernel_stats.csv' | head -1) $O/rocprof_gemv.json > /dev/null || exit 1
cp $(find $O/prof_stats -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats_mistral7b.csv
else
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 16 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 16 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/prefill_profile.py mistral-7b 512 > /dev/null 2> $O/pmc_mfma.err || exit 1
cd $R
python3 tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma_mistral_t512.json > /dev/null
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_gemv.json > /dev/null
timeout -k 10 300 python3 tools/batch_bench.py --profile --batches 1,2,4,8 > $O/batch_decode_mistral7b.txt 2> $O/batch.err || exit 1
(for t in 1 2 4 8; do timeout -k 10 200 python3 tools/tp_decode_profile.py mistral-7b $t 512; done) > $O/tp_decode_emulated.txt 2> $O/tp_decode.err || exit 1
fi
find $O -name '*kernel_trace.csv' -size +20M -delete
find $O -name '*counter_collection.csv' -size +20M -delete
ls -la $O
