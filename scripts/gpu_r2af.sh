#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
for V in 4 2; do echo "TRSM_FR=$V"; TMPC_TRSM_FR=$V timeout 300 python scripts/factor_bench.py 512,64,300 64,64,300 8,64,300; done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2af_bench.json 2> gpurun_out/r2af_bench.err; python scripts/show_bench.py gpurun_out/r2af_bench.json
