#!/bin/bash
# GPU box: L2 / fabric counters per kernel over one prefill.  usage: collect_l2_pass.sh model T [tag]
# Output: gpurun_out/l2/<tag>.json (per kernel: TCC hits / misses / fabric read requests per dispatch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
model=${1:-mistral-7b}; T=${2:-512}; tag=${3:-l2_${model}_$T}
O=$R/gpurun_out/l2
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE"; do
    n=$(echo $set | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_$n -- python3 $R/tools/prefill_profile.py $model $T 4 > /dev/null 2> $O/${tag}_$n.err || { echo "pass $n failed"; tail -3 $O/${tag}_$n.err; }
done
python3 $R/tools/pmc_table.py $O/$tag.json $O/${tag}_* > /dev/null
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -size +2M -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
cat $O/$tag.json
