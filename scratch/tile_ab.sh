#!/bin/bash
# tile_ab.sh <name> ...: tile-kernel timing (8192 tokens, 32 experts) against scratch/exp/<name>.so on one box
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 120 python scratch/tile_time.py 8192 2>&1 | grep "T=" || exit 1
done
