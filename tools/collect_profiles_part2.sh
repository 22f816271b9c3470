#!/bin/bash
# Run on the GPU box (via gpurun): bench lines + rocprofv3 kernel stats + PMC passes into gpurun_out/final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
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
# keep the merged output small: the per-dispatch traces are large
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O
