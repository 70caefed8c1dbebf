#!/bin/bash
# rocprofv3 kernel-trace summaries of bench.py (bs=32 leg and bs=1 leg separately, so each kernel row is one
# configuration) + HBM read-request passes on the MoE kernels.  usage: profile_round.sh r01
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
mkdir -p gpurun_out/prof_$tag
python3 -c "import json, bench; json.dump(bench.source_stamps(), open('gpurun_out/prof_$tag/source_blobs.json', 'w'), indent=1)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bs1 > gpurun_out/prof_${tag}_bench.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_bench.log; exit 1; }
grep '^{' gpurun_out/prof_${tag}_bench.log > gpurun_out/prof_$tag/bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace_bs1 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bs1 --batch-per-gpu 1 > gpurun_out/prof_${tag}_bench_bs1.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_bench_bs1.log; exit 1; }
grep '^{' gpurun_out/prof_${tag}_bench_bs1.log > gpurun_out/prof_$tag/bench_line_bs1.json
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/prof_$tag/pmc_rd -- python scratch/prof_moe.py 32 2 > gpurun_out/prof_${tag}_pmc.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$tag/pmc_fetch -- python scratch/prof_moe.py 32 2 > gpurun_out/prof_${tag}_pmc2.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_$tag/pmc_tile_sq -- python scratch/prof_tile.py 8192 2 > gpurun_out/prof_${tag}_pmc3.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc3.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/prof_$tag/pmc_tile_grbm -- python scratch/prof_tile.py 8192 2 > gpurun_out/prof_${tag}_pmc4.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc4.log; exit 1; }
echo done
