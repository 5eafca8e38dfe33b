#!/bin/bash
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
TAG=${1:-r2}
B0="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof/p1 -o p1 -- $B0 > /dev/null 2> /tmp/prof/p1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/prof/p2 -o p2 -- $B0 > /dev/null 2> /tmp/prof/p2.err
python3 scripts/pmc_summary.py gpurun_out/${TAG}_pmc_mem.txt /tmp/prof/p1 /tmp/prof/p2
grep -E "k_cr_(update|trsm|potrf)|k_cr_bwd|k_cr_fwd" gpurun_out/${TAG}_pmc_mem.txt
