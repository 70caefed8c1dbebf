#!/bin/bash
# bench without the side legs: headline + bs1 only.  usage: bs_quick.sh <tag>
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --prefill-tokens 0 --no-dequant-leg --no-full-layer --no-rank-of-8 > gpurun_out/q_$1.json 2> gpurun_out/q_$1.err
