#!/bin/bash
# counter passes over the prompt attention kernel; prints per-counter means
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_attn/p$i -- python scratch/prof_attn.py > gpurun_out/pmc_attn_p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_attn_p$i.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_attn/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_prefill" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):.6g}  (n={len(v)})")
PY
