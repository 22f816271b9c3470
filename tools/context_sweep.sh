for spec in "mistral-7b 512" "mistral-7b 2048" "mistral-7b 8192" "mistral-7b 16384" "qwen2-7b 512" "qwen2-7b 4096" "qwen2-7b 16384" "qwen2-7b 30000"; do
  set -- $spec
  FL_BENCH_BATCH=0 timeout -k 10 400 python3 bench.py --model $1 --prompt $2 --steps 64 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=[k for k in j['kernels'] if k['name']=='attn_decode'][0]
print('$1 S=$2: %.1f tokens/s  %.3f ms/step  attention %.1f us/layer  e2e %.0f GB/s (%.3f of 8 TB/s)  prefill %.1f ms (%.0f tokens/s)' % (j['value'], j['ms_per_step'], a['us_per_launch'], j['e2e_hbm']['achieved_GBps'], j['e2e_hbm']['frac_of_8TBps_per_gpu'], j['prefill']['ms'], j['prefill']['tokens_per_sec']))" || echo "$spec failed"
done
