#!/bin/bash
# rocprofv3 kernel-trace summaries of bench.py (one run per leg, so each kernel row is one workload) + HBM read-request passes on the MoE kernels.  usage: profile_round.sh r01
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
mkdir -p gpurun_out/prof_$tag
python3 -c "import json, bench; json.dump(bench.source_stamps(), open('gpurun_out/prof_$tag/source_blobs.json', 'w'), indent=1)"
# one trace per leg, so that a kernel row of the summary is one workload (every run also holds the headline step's own
# launches: bs=32 hot path, or bs=1 with --batch-per-gpu 1)
trace_leg() {   # trace_leg <name> <bench.py arguments...>
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace_$name -- python bench.py "$@" > gpurun_out/prof_${tag}_bench_$name.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_bench_$name.log; exit 1; }
  grep '^{' gpurun_out/prof_${tag}_bench_$name.log > gpurun_out/prof_$tag/bench_line_$name.json
}
# the same two runs without the profiler first: under rocprofv3 every launch of a graph carries the tracer's own cost, so the
# bench's HIP-event figures of the traced runs read high; the summary compares the trace rows with these plain runs
timeout -k 10 300 python bench.py --legs roofline 2> gpurun_out/prof_${tag}_plain32.err | grep '^{' > gpurun_out/prof_$tag/bench_line_plain32.json || exit 1
timeout -k 10 300 python bench.py --legs roofline --batch-per-gpu 1 2> gpurun_out/prof_${tag}_plain1.err | grep '^{' > gpurun_out/prof_$tag/bench_line_plain1.json || exit 1
trace_leg hot32 --steps 10 --warmup 3 --legs roofline
trace_leg hot1 --steps 10 --warmup 3 --legs roofline --batch-per-gpu 1
trace_leg prefill --steps 2 --warmup 1 --windows 1 --legs prefill
trace_leg full --steps 5 --warmup 2 --windows 1 --legs full
trace_leg side --steps 2 --warmup 1 --windows 1 --legs dequant,rank8
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/prof_$tag/pmc_rd -- python scratch/prof_moe.py 32 2 > gpurun_out/prof_${tag}_pmc.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$tag/pmc_fetch -- python scratch/prof_moe.py 32 2 > gpurun_out/prof_${tag}_pmc2.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_$tag/pmc_tile_sq -- python scratch/prof_tile.py 8192 2 > gpurun_out/prof_${tag}_pmc3.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc3.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/prof_$tag/pmc_tile_grbm -- python scratch/prof_tile.py 8192 2 > gpurun_out/prof_${tag}_pmc4.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_pmc4.log; exit 1; }
echo done
