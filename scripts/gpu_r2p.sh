#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "step3" 2>&1 | tail -40 > gpurun_out/r2p_step3.log
cat gpurun_out/r2p_step3.log
