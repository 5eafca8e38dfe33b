#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
timeout 900 python scripts/robustness_sweep.py 2>&1 | tail -12 | cut -c1-250
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2aa_bench.json 2> gpurun_out/r2aa_bench.err; python scripts/show_bench.py gpurun_out/r2aa_bench.json
