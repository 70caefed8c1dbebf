#!/bin/bash
# prefill_ab.sh <shape> <name> ...: the prefill leg of bench.py (tile kernel) against scratch/exp/<name>.so on one box
shape=$1; shift
for v in "$@"; do
  echo "== $v ($shape)"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 400 python bench.py --shape $shape --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-bs1 --no-roofline --no-dequant-leg --no-full-layer --no-rank-of-8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; r=p['roofline']
print('prefill %.0f tok/s  %.3f ms/layer (router+moe %.3f)  gate_up %.0f us frac %.3f  down %.0f us frac %.3f' % (p['value'], p['ms_per_layer'], p['ms_per_layer_router_and_moe'], r['avg_us'], r['frac'], r['down_kernel']['avg_us'], r['down_kernel']['frac']))" || exit 1
done
