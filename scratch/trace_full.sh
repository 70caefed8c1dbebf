#!/bin/bash
# kernel trace of the whole-decoder-stack leg (bs=32 and bs=1): per-kernel average durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_full
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_full -- python bench.py --steps 5 --warmup 2 --windows 1 --legs full > gpurun_out/trace_full.log 2>&1 || { tail -5 gpurun_out/trace_full.log; exit 1; }
python - <<'PY'
import csv, glob, collections
g = collections.defaultdict(list)
for f in glob.glob("gpurun_out/trace_full/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "m355" not in n: continue
        key = (n.split("(")[0][:70], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
        g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(g.items()):
    print(f"{k[0]:72s} wg={k[1]:5d} n={len(v):5d} avg={sum(v)/len(v)/1e3:8.2f} us")
PY
