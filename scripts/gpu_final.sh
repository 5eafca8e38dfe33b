#!/bin/bash
# the evidence set of a round at HEAD (gpurun): GPU suite log, profile set, default bench line, parity fuzz at two seeds
export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r6_final}
timeout 2400 python -m pytest tests/ -m gpu -q -p no:cacheprovider -s < /dev/null 2>&1 | grep -E "passed|failed|FAILED|worst|back-offs|AWE shape|Step 3 p=|c5 share|c3 shape|bench shape|Error" | cut -c 1-330 > gpurun_out/${TAG}_gpu_tests.log; tail -n 2 gpurun_out/${TAG}_gpu_tests.log
bash scripts/gpu_prof.sh $TAG > gpurun_out/${TAG}_prof_stdout.txt 2>&1; head -n 8 gpurun_out/${TAG}_kernel_stats.txt | cut -c 1-150
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python scripts/show_bench.py gpurun_out/${TAG}_bench.json | cut -c 1-250
timeout 900 python tests/tools/parity_fuzz.py 120 90001 2>&1 | tail -n 5 | cut -c 1-600; cp gpurun_out/parity_fuzz.json gpurun_out/${TAG}_fuzz_seed90001.json
timeout 900 python tests/tools/parity_fuzz.py 40 90002 12 24 2>&1 | tail -n 3 | cut -c 1-600; cp gpurun_out/parity_fuzz.json gpurun_out/${TAG}_fuzz_seed90002_nx24.json
timeout 900 python tests/tools/tight_models_fuzz.py 97002 150 2>&1 | tail -n 1 | cut -c 1-800 > gpurun_out/${TAG}_tight_models_fuzz_seed97002.json; cat gpurun_out/${TAG}_tight_models_fuzz_seed97002.json
