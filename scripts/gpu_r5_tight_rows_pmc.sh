#!/bin/bash
# round 5: vector-ALU counters of the double-double kernels under the tight mode on the Step 2 model (one pass; program directly after --)
mkdir -p gpurun_out /tmp/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof/tr1 -o tr1 -- python3 scripts/tight_rows_timing.py 64 > gpurun_out/r5_tight_rows_pmc_stdout.txt 2> /tmp/prof/tr1.err
timeout 120 python3 scripts/pmc_summary.py gpurun_out/r5_tight_rows_pmc.txt /tmp/prof/tr1 < /dev/null
tail -n 2 gpurun_out/r5_tight_rows_pmc_stdout.txt; grep -E "k_dd_(update|trsm|potrf|aug_fill|solve_border)|k_polish" gpurun_out/r5_tight_rows_pmc.txt | cut -c1-200 | head -40
