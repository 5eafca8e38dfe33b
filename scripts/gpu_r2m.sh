#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "dual_certificate or block_cyclic" 2>&1 | tail -4
for RS in 64 128 192 320; do
  echo "rs $RS"; TMPC_CR_RS=$RS timeout 300 python scripts/factor_bench.py 512,64,300
done 2>&1 | tee gpurun_out/r2m_rs_sweep.txt
