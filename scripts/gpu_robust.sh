#!/bin/bash
# robustness sweep + traces of two hard members (gpurun)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python scripts/robustness_sweep.py 2>&1 | tail -n 14
timeout 300 python scripts/robust_trace.py 30 4 1 10.0 5 0.5 2>&1 | cut -c 1-170 > gpurun_out/robust_trace_a.txt
timeout 300 python scripts/robust_trace.py 30 4 1 100.0 3 0.5 2>&1 | cut -c 1-170 > gpurun_out/robust_trace_b.txt
