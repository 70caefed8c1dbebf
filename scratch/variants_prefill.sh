#!/bin/bash
# variants_prefill.sh <name> ...: the prompt leg of bench.py against scratch/exp/<name>.so, one after the other on the same box
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --legs prefill 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; r=p['roofline']
print('prefill %.0f tok/s ms/layer %.3f router+moe %.3f | tile gate_up %.1f us (%.3f) down %.1f us (%.3f)' % (p['value'], p['ms_per_layer'], p['ms_per_layer_router_and_moe'], r['avg_us'], r['frac'], r['down_kernel']['avg_us'], r['down_kernel']['frac']))" || exit 1
done
