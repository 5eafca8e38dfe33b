#!/bin/bash
# round 2, first GPU session: parity of the cyclic-reduction kernels, micro-benchmark, isolated timing, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r2d_tests.log
echo "tests rc=$?" >> gpurun_out/r2d_tests.log

timeout 600 python scripts/factor_bench.py > gpurun_out/r2d_factor_bench.txt 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
tail -5 gpurun_out/r2d_tests.log; cat gpurun_out/r2d_factor_bench.txt; cat gpurun_out/r2d_bench.json | cut -c1-1500
