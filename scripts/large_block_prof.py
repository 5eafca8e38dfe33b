"""The two large-block legs of bench.py (n = 48: plain model, Step 2 with 24 + 24 rows) alone, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier

which = sys.argv[1] if len(sys.argv) > 1 else 'plain'
rng = np.random.default_rng(7)
seed, nb, p, nx, mb, ng, nc = (200200, 32, 16, 40, 8, 0, 0) if which == 'plain' else (200300, 16, 8, 36, 12, 24, 24)
if len(sys.argv) > 2:
    nb = int(sys.argv[2])
A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb)
h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc)
if ng or nc:
    J = rng.standard_normal((nb, p, ng + nc, nx + mb)); ncnt = np.full((nb, p), nc, np.int32)
    run = lambda: h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
else:
    run = lambda: h.convexify_batch(A, B, H)
run()
t0 = time.perf_counter(); o = run(); t = time.perf_counter() - t0
print(f'{which}: {1e3 * t:.1f} ms, {nb * p / t:.1f} stage-conv/s, iterations {o["iters"].mean():.1f}, optimal {(o["status"] == 0).sum()} / {nb}')
h.close()
