"""Batched timing of the GEMM cores: private operands (HBM-realistic) vs shared operands (cache-resident ceiling)."""
import sys, json
import numpy as np
sys.path.insert(0, '.')
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(1, 2, 4, 1)
res = {}
cases = [(304, 304, 304, 0, 0, 'full v0'), (304, 304, 304, 1, 0, 'full v1'), (304, 304, 304, 1, 1, 'syrk v1'),
         (304, 64, 256, 0, 0, 'panel v0'), (304, 64, 256, 2, 0, 'panel v2')]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[5] in sys.argv[1:]]
for nb in (256, 512):
    for shared in (0, 16):
        for (M, N, K, var, tri, tag) in cases:
            reps = 8
            ms = h.debug_gemm_bench(nb, M, N, K, var, tri | shared, reps, 3)
            fl = 2.0 * M * N * K * (0.5 if tri == 1 else 1.0) * reps * nb
            key = f'{tag} nb={nb} {"shared" if shared else "private"}'
            res[key] = dict(ms=ms, tflops=fl / ms / 1e9)
            print(f'{key:34s}: {ms:8.3f} ms  {fl / ms / 1e9:6.2f} TF/s (algorithmic)')
json.dump(res, open('gpurun_out/gemm2_bench.json', 'w'), indent=1)
# timing experiments: where do the cycles of a slab go
for (mode, tag) in [(2, 'complete'), (4, 'no DMA in loop'), (5, 'no DMA, no LDS fetch')]:
    M = N = K = 304; reps = 8; nb = 256
    ms = h.debug_gemm_bench(nb, M, N, K, 1, 16 | (mode << 8), reps, 3)
    print(f'160x160 core, shared operands, {tag:22s}: {ms:7.3f} ms  {2.0 * 320 * 320 * 304 * reps * nb / ms / 1e9:6.2f} TF/s executed')
