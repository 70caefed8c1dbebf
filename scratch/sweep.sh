#!/bin/bash
for v in "$@"; do echo "== $v"; M355_LIB_PATH=$PWD/scratch/exp/$v.so timeout -k 10 200 python scratch/sweep.py 32 || exit 1; done
