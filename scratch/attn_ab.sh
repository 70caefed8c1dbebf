#!/bin/bash
# attn_ab.sh v1 v2 ...: prompt attention timing (8192 tokens) for each variant library in scratch/exp
for v in "$@"; do
  echo "== $v"
  M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 120 python scratch/attn_time.py 2>&1 | grep "L=8192" || exit 1
done
