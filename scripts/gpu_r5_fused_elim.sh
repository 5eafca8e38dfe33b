#!/bin/bash
# round 5, experiment (b'): block Cholesky + triangular solves of a node fused into one workgroup (k_cr_elim_dma) against the three-kernel form
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r5_fused_elim.txt
echo "# scripts/factor_bench.py, nb,p,d = 512,64,300 / 433,64,300 / 64,64,300; base = k_cr_potrf_dma + k_cr_trsm_dma + k_cr_update_dma per level; fused = k_cr_elim_dma + k_cr_update_dma" > $OUT
for rep in 1 2; do
  echo "## base (run $rep)" >> $OUT;  timeout 300 python scripts/factor_bench.py 512,64,300 433,64,300 64,64,300 >> $OUT 2>&1
  echo "## fused (run $rep)" >> $OUT; FB_FUSED_ELIM=1 timeout 300 python scripts/factor_bench.py 512,64,300 433,64,300 64,64,300 >> $OUT 2>&1
done
cat $OUT
