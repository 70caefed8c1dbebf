#!/bin/bash
# usage: pmc.sh <tag> <B> ; runs several counter passes over scratch/prof_moe.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; B=$2
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python scratch/prof_moe.py $B 2 > gpurun_out/pmc_${tag}_p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_p$i.log; exit 1; }
done
echo ok
