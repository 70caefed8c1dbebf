#!/bin/bash
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 120 python scratch/dense_time.py 2>&1 | grep "N=" || exit 1
done
