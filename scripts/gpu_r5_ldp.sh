#!/bin/bash
# round 5, experiment (c): leading dimension of the 64 x 64 tile images of the block Cholesky (LDS bank conflicts of k_cr_potrf_dma: 22.5 % at 65)
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
OUT=gpurun_out/r5_potrf_ldp.txt
echo "# scripts/factor_bench.py 433,64,300 under rocprofv3 --kernel-trace: time of k_cr_potrf_dma per factorisation for the tile-image leading dimensions 65 (product), 66, 68, 69" > $OUT
for v in 65 66 68 69 65; do
  L=$PWD/tunempc_amd/lib/libtunempc_hip_ldp$v.so; [ $v = 65 ] && L=$PWD/tunempc_amd/lib/libtunempc_hip.so
  rm -rf /tmp/prof/ldp
  TMPC_LIB=$L rocprofv3 --kernel-trace -d /tmp/prof/ldp -o kt -- python3 scripts/factor_bench.py 433,64,300 > /tmp/prof/ldp.out 2> /tmp/prof/ldp.err
  echo "## LDP = $v" >> $OUT; tail -1 /tmp/prof/ldp.out >> $OUT
  python3 scripts/rocpd_stats.py $(find /tmp/prof/ldp -name '*.db' | head -1) 2>&1 | grep "potrf\|trsm_dma\|update_dma" | cut -c1-140 >> $OUT
done
cat $OUT
