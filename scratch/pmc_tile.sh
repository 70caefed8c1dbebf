#!/bin/bash
# usage: pmc_tile.sh <tag> <T> ; counter passes over scratch/prof_tile.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; T=$2
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python scratch/prof_tile.py $T 2 > gpurun_out/pmc_${tag}_p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_p$i.log; exit 1; }
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc_${tag}/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'moe_tile' not in k: continue
        name='gate_up' if 'ILi0E' in k else 'down'
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name,cs in agg.items():
    print(name)
    for c,v in cs.items():
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
