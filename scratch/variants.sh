#!/bin/bash
# variants.sh <name> ...: bench.py against scratch/exp/<name>.so, one after the other on the same box
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --prefill-tokens 0 --no-dequant-leg --no-full-layer 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['bs1']['roofline_gate_up']
print('bs32 ms/step %.3f gate_up %.1f us down %.1f us | bs1 ms/step %.3f gate_up %.1f down %.1f' % (d['ms_per_step'], r['avg_us'], r['down_kernel']['avg_us'], d['bs1']['ms_per_step'], b['avg_us'], b['down_avg_us']))" || exit 1
done
