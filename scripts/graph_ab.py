"""A/B of TMPC_TUNE_GRAPH (one IPM iteration replayed as a captured hipGraph) on launch-bound shapes: median solve time and bit-identity of the outputs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier

for name, (seed, nb, p, nx, mb) in {'configs[1] unicycle-shaped p=30 n=5 batch=1': (200000, 1, 30, 4, 1), 'lqr-shaped p=1 n=3 batch=1': (1, 1, 1, 2, 1), 'p=30 n=5 batch=8': (200001, 8, 30, 4, 1),
                                    'configs[2] evaporation-shaped p=50 n=4 batch=256': (200100, 256, 50, 2, 2), 'p=20 n=12 (nx=10: blocks of 55) batch=4': (200400, 4, 20, 10, 2)}.items():
    A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb)
    res = {}
    for g in (0, 1):
        h = HipConvexifier(p, nx, mb, chunk=nb)
        h.set_tuning(graph=g)
        h.convexify_batch(A, B, H); h.convexify_batch(A, B, H)
        ts = []
        for _ in range(11):
            t0 = time.perf_counter(); o = h.convexify_batch(A, B, H); ts.append(time.perf_counter() - t0)
        h.close()
        res[g] = (float(np.median(ts)), o)
    same = all(np.array_equal(res[0][1][k], res[1][1][k]) for k in ('Hc', 'P', 'kappa', 'status', 'iters'))
    print(f'{name:52s} plain launches {1e3 * res[0][0]:8.3f} ms   graph {1e3 * res[1][0]:8.3f} ms   x{res[0][0] / res[1][0]:.2f}   iterations {res[1][1]["iters"].mean():.1f}   outputs bit-identical: {same}', flush=True)
