#!/bin/bash
# Round 5, GPU box (via gpurun): the multi-rank rehearsals a one-GPU box allows, at the commit in .fl_commit.
#   8 ranks (4 processes x 2 rank threads) at full Mistral-7B shapes, one-shot kernels and the exchange fused in the GEMV epilogues;
#   bench.py --gpus 4 with every rank on this GPU: Mistral-7B 512 / 32 (with the group's batched_decode_32 leg) and Qwen2-7B 4096 / 32.
# All ranks share one HBM: none of these is a scaling figure.  Output: gpurun_out/r05/ (copied to profiles/r05/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout -k 10 500 python3 tools/tp_rehearsal.py --ranks 8 --procs 4 --model mistral-7b --prompt 512 --steps 64 --out $O/rehearsal_tp8_mistral7b.json > $O/rehearsal_tp8.log 2>&1 || { echo "tp8 rehearsal failed"; tail -5 $O/rehearsal_tp8.log; exit 1; }
timeout -k 10 500 python3 tools/tp_rehearsal.py --ranks 8 --procs 4 --model mistral-7b --prompt 512 --steps 64 --fused --out $O/rehearsal_tp8_mistral7b_fused.json > $O/rehearsal_tp8_fused.log 2>&1 || { echo "tp8 fused rehearsal failed"; tail -5 $O/rehearsal_tp8_fused.log; exit 1; }
FL_BENCH_SAME_DEVICE=1 timeout -k 10 500 python3 bench.py --gpus 4 --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_tp4_mistral_same_device.json 2> $O/bench_tp4_mistral.err || { echo "bench tp4 mistral failed"; tail -5 $O/bench_tp4_mistral.err; exit 1; }
FL_BENCH_SAME_DEVICE=1 timeout -k 10 500 python3 bench.py --gpus 4 --model qwen2-7b --prompt 4096 --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_tp4_qwen2_4k_same_device.json 2> $O/bench_tp4_qwen2.err || { echo "bench tp4 qwen2 failed"; tail -5 $O/bench_tp4_qwen2.err; exit 1; }
python3 - <<PY
import json
for f in ("rehearsal_tp8_mistral7b.json", "rehearsal_tp8_mistral7b_fused.json", "bench_tp4_mistral_same_device.json", "bench_tp4_qwen2_4k_same_device.json"):
    d = json.load(open("$O/" + f))
    print(f, {k: d.get(k) for k in ("ranks_agree", "tokens_equal_emulated", "tp_vs_single_gpu_rel_l2", "value", "ms_per_step", "ms_per_step_shared_gpu")},
          (d.get("batched_decode_32") or {}).get("aggregate_tokens_per_sec"), (d.get("config") or {}).get("tp_fallback_level"))
PY
