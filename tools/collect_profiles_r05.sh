#!/bin/bash
# Round 5, GPU box (via gpurun).  part "prefill": kernel stats + PMC passes (separate runs) of one Qwen2-7B 4096-token prefill and one
# Mistral-7B 512-token prefill.  part "bench": the bench lines + rocprofv3 kernel stats + HBM traffic passes of the default bench.
# Output: gpurun_out/r05/ (copied to profiles/r05/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
pass() {   # name, counters..., then "--", then the python arguments
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    timeout -k 10 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/$name -- python3 "$@" > /dev/null 2> $O/$name.err || { echo "pass $name failed"; tail -3 $O/$name.err; return 1; }
}
if [ "$1" = "prefill" ]; then
for spec in "qwen2_4k qwen2-7b 4096" "mistral_t512 mistral-7b 512"; do
    set -- $spec; tag=$1; model=$2; T=$3
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/tools/prefill_profile.py $model $T > $O/prefill_profile_$tag.txt 2> $O/stats_$tag.err || exit 1
    f=$(find $O/stats_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] || { echo "no kernel stats for $tag"; exit 1; }
    cp "$f" $O/rocprofv3_kernel_stats_$tag.csv
    # the stats must describe the kernels the same command's own listing names (a stale attention kernel in a committed file)
    python3 $R/tools/check_profile_kernels.py $O/rocprofv3_kernel_stats_$tag.csv $O/prefill_profile_$tag.txt --stamp || exit 1
    pass mfma_$tag SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $R/tools/prefill_profile.py $model $T || exit 1
    pass lds_$tag SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -- $R/tools/prefill_profile.py $model $T || echo "(lds pass skipped)"
    pass wave_$tag SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $R/tools/prefill_profile.py $model $T || echo "(wave pass skipped)"
    python3 $R/tools/pmc_table.py $O/pmc_mfma_$tag.json $O/mfma_$tag $O/lds_$tag $O/wave_$tag > /dev/null || exit 1
done
else
cd $R
if [ "$1" != "traffic" ]; then
timeout -k 10 600 python3 bench.py > $O/bench_mistral7b_decode.json 2> $O/bench_mistral7b.err || exit 1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 64 --no-cpu-baseline --no-secondary --no-traffic > $O/prof_stats_bench.json 2> $O/prof_stats.err || exit 1
f=$(find $O/prof_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] || { echo "no kernel stats"; exit 1; }
cp "$f" $O/rocprofv3_kernel_stats_mistral7b.csv
python3 $R/tools/check_profile_kernels.py $O/rocprofv3_kernel_stats_mistral7b.csv $O/prof_stats_bench.json --stamp || exit 1
python3 $R/tools/check_profile_kernels.py $O/rocprofv3_kernel_stats_mistral7b.csv $O/bench_mistral7b_decode.json || exit 1
python3 $R/tools/rocprof_gemv.py "$f" $O/rocprof_gemv.json > /dev/null || exit 1
fi
cd /tmp
# (counter passes without the batched-decode legs, as bench.py's own in-run passes: the 8- and 32-stream legs add ~40k dispatches and the
# profiler's counter collection segfaulted behind them; the counters are read for gemv_kernel only)
export FL_BENCH_BATCH=0
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 16 --no-cpu-baseline --no-secondary --no-traffic > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 16 --no-cpu-baseline --no-secondary --no-traffic > /dev/null 2> $O/pmc_write.err || exit 1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_gemv.json > /dev/null || exit 1
fi
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*counter_collection.csv' -size +4M -delete
find $O -name "*.db" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O
