#!/bin/bash
# full_ab.sh <name> ...: whole decoder stack (bench.py --legs full) against scratch/exp/<name>.so
for v in "$@"; do
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --legs full 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['full_decoder_stack']
print('$v', [(x.get('batch'), round(x.get('ms_per_step'),4)) for x in f])" || exit 1
done
