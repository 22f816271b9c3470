#!/bin/bash
# sweep decode-attention geometry on the Mistral-7B bench (GPU box)
for nw in 4 16; do for ns in 1 2 4 8 13 26 52; do
FL_ATTN_NW=$nw FL_ATTN_NSPLIT=$ns timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('nw=$nw nsplit=$ns',d['value'],[(k['name'],k['us_per_step']) for k in d['kernels'] if 'attn' in k['name']])"
done; done
