#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for CH in 0 10 30; do
echo "CHORD=$CH"
TMPC_CHORD=$CH timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
TMPC_CHORD=$CH timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2z_bench_$CH.json 2> gpurun_out/r2z_bench.err; python scripts/show_bench.py gpurun_out/r2z_bench_$CH.json
done
