#!/bin/bash
# A/B of two builds of the library on one box: gpu_ab_lib.sh BASE.so  (the in-tree library against BASE.so: bench twice each; the order
# flips in the second pair -- the run that comes second finds the part warmer and its MFMA kernels ~0.3 % slower)
mkdir -p gpurun_out; export TMPDIR=/tmp
BASE=$1
run_base() { TMPC_LIB=$BASE timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_base.json 2> gpurun_out/ab_base.err; echo -n "base: "; python scripts/show_bench.py gpurun_out/ab_base.json | cut -c1-110; }
run_new() { timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err; echo -n "new:  "; python scripts/show_bench.py gpurun_out/ab_new.json | cut -c1-110; }
run_base; run_new; run_new; run_base
