#!/bin/bash
# full GPU suite + parity fuzz (five models) + robustness sweep + A/B-free bench (gpurun)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|rror" | head -n 8 | tee gpurun_out/gpu_tests.log
timeout 900 python tests/tools/parity_fuzz.py ${FUZZ_ARGS:-} 2>&1 | tail -n 4
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err; python scripts/show_bench.py gpurun_out/fc_bench.json | cut -c 1-200
