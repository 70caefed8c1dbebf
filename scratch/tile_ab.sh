#!/bin/bash
# tile_ab.sh v1 v2 ...: gate_up / down time of the tile kernel at T=8192 for each variant library in scratch/exp
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 120 python scratch/tile_time.py 8192 2>&1 | grep -v "all:" || exit 1
done
