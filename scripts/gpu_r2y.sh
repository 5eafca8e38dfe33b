#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head
for V in 1 2; do echo "TRSM_RR=$V"; TMPC_TRSM_RR=$V timeout 300 python scripts/factor_bench.py 512,64,300 64,64,300 8,64,300; done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2y_bench.json 2> gpurun_out/r2y_bench.err; python scripts/show_bench.py gpurun_out/r2y_bench.json
