#!/bin/bash
# kernel trace of the bench workload (program directly after --)
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
TAG=${1:-r2}
rocprofv3 --kernel-trace -d /tmp/prof/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_prof_bench.json 2> /tmp/prof/kt.err
python3 scripts/rocpd_stats.py $(find /tmp/prof/kt -name '*.db' | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -45 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
