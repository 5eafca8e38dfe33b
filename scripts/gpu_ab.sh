#!/bin/bash
# A/B of one environment switch: gpu_ab.sh VAR v1 v2 ...  (factor bench per value, then the full GPU suite and a bench with the LAST value)
mkdir -p gpurun_out; export TMPDIR=/tmp
VAR=$1; shift
for V in "$@"; do echo "$VAR=$V"; env $VAR=$V timeout 300 python scripts/factor_bench.py 512,64,300 64,64,300; LAST=$V; done
env $VAR=$LAST timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
env $VAR=$LAST timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_bench.json 2> gpurun_out/ab_bench.err; python scripts/show_bench.py gpurun_out/ab_bench.json
