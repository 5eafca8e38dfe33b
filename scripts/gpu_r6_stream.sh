#!/bin/bash
# round 6, experiment (a): the update kernel as one slab stream over T consecutive tiles per workgroup (k_cr_update_dma_stream) against one tile per workgroup
export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/r6_update_stream_windowed_static_ring.txt
echo "# scripts/factor_bench.py, nb,p,d = 433,64,300 / 512,64,300 / 64,64,300; FB_UPDATE_STREAM = tiles per workgroup (+300: ring of three slab buffers); 0 = k_cr_update_dma" > $OUT
python - >> $OUT 2>&1 <<'PY'
import sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
# bit-identity first: a batch at the bench stage shape (and one with blocks of 210, one of 48) through both kernels
for (nb, p, nx, mb) in ((6, 64, 24, 8), (3, 20, 20, 10), (4, 9, 9, 6), (5, 7, 12, 4)):
    A, B, H = synthetic.gen_batch(31337, nb, p, nx, mb)
    dig = {}
    for v in (0, 2, 5, 8, 13):
        h = HipConvexifier(p, nx, mb, chunk=nb); h.set_tuning(update_stream=v)
        o = h.convexify_batch(A, B, H); h.close()
        dig[v] = hashlib.sha256(o['Hc'].tobytes() + o['P'].tobytes() + o['iters'].tobytes()).hexdigest()[:16]
    print('bit-identity', (nb, p, nx, mb), dig, 'OK' if len(set(dig.values())) == 1 else 'MISMATCH', flush=True)
PY
for rep in 1; do
  for v in 0 2 4 8 16 0; do
    echo "## FB_UPDATE_STREAM=$v (run $rep)" >> $OUT; FB_UPDATE_STREAM=$v timeout 300 python scripts/factor_bench.py 433,64,300 512,64,300 64,64,300 >> $OUT 2>&1
  done
done
cat $OUT | cut -c 1-200
