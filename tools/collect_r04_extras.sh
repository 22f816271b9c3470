#!/bin/bash
# Round 4 extras (GPU box): in-kernel stamps of the 128 x 256 kernel, the one-process A/B lines, the experimental library's tests.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
rm -f $O/h4_stamps.txt
FL_H4_STAMPS=$O/h4_stamps.txt timeout -k 10 200 python3 tools/prefill_profile.py mistral-7b 512 1 > /dev/null 2>&1
python3 tools/stamps_h4.py $O/h4_stamps.txt > $O/h4_stamps_summary.txt; rm -f $O/h4_stamps.txt
(timeout -k 10 400 python3 tools/tune_ab.py mistral-7b gemm_h4 0 1 300 384 512 640 768 1024; timeout -k 10 300 python3 tools/tune_ab.py qwen2-7b gemm_h4 0 1 384 512 640) 2>/dev/null > $O/ab_gemm_h4.txt
(timeout -k 10 300 python3 tools/tune_ab.py qwen2-7b h4_tail 0 1 384 512 4096; timeout -k 10 300 python3 tools/tune_ab.py mistral-7b rs_lazy 0 1 384 512 640) 2>/dev/null > $O/ab_h4_tail_rs_lazy.txt
(echo "== default library: pytest -m gpu tests/test_gpu_h4_shared_gpu.py tests/test_gpu_ops.py -k h4"; timeout -k 10 400 python3 -m pytest -q -m gpu tests/test_gpu_h4_shared_gpu.py tests/test_gpu_ops.py -k "h4 or shared" 2>&1 | tail -3;
 echo "== experimental library (FL_LIB_PATH=fastllm_amd/lib/libfastllm_mi355x_exp.so): the kernels the default build leaves out";
 FL_LIB_PATH=$R/fastllm_amd/lib/libfastllm_mi355x_exp.so timeout -k 10 300 python3 -m pytest -q -m gpu tests/test_gpu_engine.py tests/test_gpu_attn_prefetch.py tests/test_gpu_parity.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -k "engine or prefetch or fused_attention_oproj or skinny or variant_paths" 2>&1 | tail -3) > $O/gputest_default_and_experimental.txt
ls -la $O | head -40
