#!/bin/bash
# ab_bs.sh "<batch sizes>" v1 v2 ...: decode step and gate_up / down times for each variant library in scratch/exp
bss=$1; shift
for v in "$@"; do
  for b in $bss; do
    M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 200 python bench.py --batch-per-gpu $b --no-bs1 --no-cpu-baseline --prefill-tokens 0 --steps 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v bs=$b ms/step %.4f gate_up %.1f us down %.1f us' % (d['ms_per_step'], r['avg_us'], r['down_kernel']['avg_us']))" || exit 1
  done
done
