#!/bin/bash
# trace_ab.sh <name> ...: kernel-trace statistics of the bs=32 hot path against scratch/exp/<name>.so (top rows)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export M355_LIB_PATH=$GRAFT_REPO_ROOT/scratch/exp/$v.so
  rm -rf gpurun_out/trace_ab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_ab_$v -- python bench.py --steps 10 --warmup 3 --legs none ${BENCH_EXTRA} > gpurun_out/trace_ab_$v.log 2>&1 || { tail -5 gpurun_out/trace_ab_$v.log; exit 1; }
  f=$(find gpurun_out/trace_ab_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("TRACE_ROWS", "8"))]:
    print('%-90s n=%6s avg %8.2f us  min %8.2f  max %8.2f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
