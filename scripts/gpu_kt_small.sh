#!/bin/bash
# kernel trace of a small configuration (unicycle-shaped, batch 1): where does a latency-bound solve spend its time?
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof/kts -o kts -- python3 scripts/config_sweep.py c2 > gpurun_out/small_kt_stdout.txt 2> /tmp/prof/kts.err
python3 scripts/rocpd_stats.py $(find /tmp/prof/kts -name '*.db' | head -1) > gpurun_out/small_kernel_stats.txt 2>&1
head -n 40 gpurun_out/small_kernel_stats.txt | cut -c1-150
