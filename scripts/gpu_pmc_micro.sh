#!/bin/bash
# SQ counters of the GEMM-core micro-benchmark (program directly after --)
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
cd scripts/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 dma_gemm.hip -o /tmp/dma_gemm 2>&1 | grep error; cd ../..
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof/m3 -o m3 -- /tmp/dma_gemm 4096 > /tmp/prof/m3.out 2> /tmp/prof/m3.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/prof/m4 -o m4 -- /tmp/dma_gemm 4096 > /tmp/prof/m4.out 2> /tmp/prof/m4.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d /tmp/prof/m5 -o m5 -- /tmp/dma_gemm 4096 > /tmp/prof/m5.out 2> /tmp/prof/m5.err
python3 scripts/pmc_summary.py gpurun_out/r2_pmc_micro.txt /tmp/prof/m3 /tmp/prof/m4 /tmp/prof/m5
tail -3 /tmp/prof/m3.err /tmp/prof/m5.err | cut -c1-200
cat /tmp/prof/m3.out
cut -c1-200 gpurun_out/r2_pmc_micro.txt | head -150
