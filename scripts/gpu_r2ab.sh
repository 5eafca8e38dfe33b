#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
for NT in 64 256; do
TMPC_STAGE_NT=$NT timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2ab_bench_$NT.json 2> gpurun_out/r2ab_bench.err; echo "STAGE_NT=$NT"; python scripts/show_bench.py gpurun_out/r2ab_bench_$NT.json
done
