#!/bin/bash
# variants_bs.sh <batch> <name> ...: hot-path step time (no side legs) at one batch size against scratch/exp/<name>.so
b=$1; shift
for v in "$@"; do
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --legs none --batch-per-gpu $b 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v bs=$b ms/step %.4f' % d['ms_per_step'])" || exit 1
done
