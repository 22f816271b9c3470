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
# next-row components and the multi-rank rehearsal (all ranks on this one GPU: protocol check, not a scaling figure)
timeout -k 10 300 python3 tools/batch_bench.py --profile --batches 1,2,3,4,6,8 > $O/batch_decode_mistral7b.txt 2> $O/batch.err || exit 1
timeout -k 10 200 python3 tools/sample_cost.py > $O/sample_cost.txt 2> $O/sample.err || exit 1
timeout -k 10 200 python3 tools/gemm_probe.py > $O/gemm_probe.txt 2> $O/gemm_probe.err || exit 1
(timeout -k 10 200 python3 tools/prefill_sweep.py mistral-7b; timeout -k 10 200 python3 tools/prefill_sweep.py tinyllama-1.1b) 2> $O/prefill_sweep.err | grep 'T=' > $O/prefill_sweep.txt || exit 1
# prompt lengths that are not round numbers (tile / block quantisation), Mistral-7B, Qwen2-7B, TinyLlama
(timeout -k 10 300 python3 tools/prefill_ragged.py mistral-7b; timeout -k 10 300 python3 tools/prefill_ragged.py qwen2-7b; timeout -k 10 200 python3 tools/prefill_ragged.py tinyllama-1.1b 256 384 512 640 768 1024 1100 1536 2048) 2> $O/prefill_ragged.err | grep 'T=' > $O/prefill_ragged.txt || exit 1
# the way the driver invokes the N-GPU bench: no launcher (bench.py starts its ranks itself)
for n in 2 4; do
  FL_BENCH_SAME_DEVICE=1 FL_BENCH_BATCH=0 timeout -k 10 400 python3 bench.py --gpus $n --steps 64 --warmup 8 --no-cpu-baseline > $O/bench_tp${n}_same_device.json 2> $O/bench_tp${n}.err || exit 1
done
timeout -k 10 300 python3 tools/skinny_probe.py 1,8,128 > $O/projection_probe_cold.txt 2> $O/skinny.err || exit 1
timeout -k 10 500 python3 tools/cpu_baseline_full.py > $O/cpu_baseline_full_depth.txt 2> $O/cpu_full.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/prefill_profile.py mistral-7b 512 > /dev/null 2> $O/pmc_mfma.err || exit 1
cd $R
python3 tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma_mistral_t512.json > /dev/null
KS=$(ls $O/prof_stats/*/*kernel_stats.csv $O/prof_stats/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$KS" ] || { echo "no kernel_stats.csv under $O/prof_stats: rocprof_gemv.json NOT refreshed"; exit 1; }
python3 tools/rocprof_gemv.py "$KS" $O/rocprof_gemv.json > /dev/null || exit 1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_gemv.json > /dev/null
# keep the merged output small: the per-dispatch traces are large
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O
