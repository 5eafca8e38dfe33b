#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/persist_check.py $1 > gpurun_out/r5_persist_check.txt 2>&1; cat gpurun_out/r5_persist_check.txt | cut -c1-260
timeout 600 python tests/tools/result_digest.py > gpurun_out/r5_digest.txt 2>&1; cat gpurun_out/r5_digest.txt
