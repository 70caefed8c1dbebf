#!/bin/bash
# tile_ab.sh "<T list>" v1 v2 ...: tile kernel times for each variant library in scratch/exp
ts=$1; shift
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 120 python scratch/tile_time.py $ts 2>&1 | grep "tile" | grep -v "all:" || exit 1
done
