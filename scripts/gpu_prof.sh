#!/bin/bash
# kernel trace + PMC passes of the bench workload (each rocprofv3 run is its own process; program directly after --)
mkdir -p gpurun_out /tmp/prof
export TMPDIR=/tmp
TAG=${1:-r2}
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace -d /tmp/prof/kt -o kt -- $B > gpurun_out/${TAG}_prof_bench.json 2> /tmp/prof/kt.err
python3 scripts/rocpd_stats.py $(find /tmp/prof/kt -name '*.db' | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
B0="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof/p1 -o p1 -- $B0 > /dev/null 2> /tmp/prof/p1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/prof/p2 -o p2 -- $B0 > /dev/null 2> /tmp/prof/p2.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof/p3 -o p3 -- $B0 > /dev/null 2> /tmp/prof/p3.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/prof/p4 -o p4 -- $B0 > /dev/null 2> /tmp/prof/p4.err
python3 scripts/pmc_summary.py gpurun_out/${TAG}_pmc.txt /tmp/prof/p1 /tmp/prof/p2 /tmp/prof/p3 /tmp/prof/p4
mkdir -p gpurun_out/profiles_out; python3 scripts/pmc_traffic.py /tmp/prof/p1 /tmp/prof/p2 && cp profiles/r6_traffic.json gpurun_out/${TAG}_traffic.json
tail -n 3 /tmp/prof/p1.err /tmp/prof/p2.err /tmp/prof/p3.err /tmp/prof/p4.err > gpurun_out/${TAG}_prof_err.txt 2>&1
head -n 25 gpurun_out/${TAG}_kernel_stats.txt; grep -E "k_cr_(update|trsm|potrf)" gpurun_out/${TAG}_pmc.txt | cut -c1-190 | head -80
