#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head
timeout 900 python scripts/chord_probe.py 128 > gpurun_out/r2q_chord.txt 2>&1
cut -c1-260 gpurun_out/r2q_chord.txt | grep chord_step
