#!/bin/bash
# dq_ab.sh <name> ...: the standalone dequant leg of bench.py against scratch/exp/<name>.so on one box
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-bs1 --no-roofline --prefill-tokens 0 --no-full-layer --no-rank-of-8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['dequant']; print('dequant %.1f us  %.0f GB/s  frac %.3f' % (d['avg_us'], d['achieved'], d['frac']))" || exit 1
done
