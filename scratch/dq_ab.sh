#!/bin/bash
# dq_ab.sh v1 v2 ...: the standalone dequantiser's time for each variant library in scratch/exp
for v in "$@"; do
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 200 python bench.py --legs dequant --steps 3 --windows 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['dequant']
print('$v %.1f us frac %.3f' % (d['avg_us'], d['frac']))" || exit 1
done
