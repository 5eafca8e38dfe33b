"""Latency of single small problems (median of nine solves after a warm-up): AWE real size, batches of 1 and 8; LQR-sized p = 1."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
for (nb, p, nx, mb) in [(1, 40, 9, 6), (8, 40, 9, 6), (1, 1, 3, 1), (1, 1, 4, 2)]:
    A, B, H = synthetic.gen_batch(500000, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb)
    h.convexify_batch(A, B, H)
    ts = []
    for _ in range(9):
        t = time.perf_counter(); o = h.convexify_batch(A, B, H); ts.append(time.perf_counter() - t)
    print(nb, p, nx, mb, 'median %.2f ms min %.2f ms iters %d status %s' % (1e3 * np.median(ts), 1e3 * min(ts), o['iters'].max(), o['status'].tolist()[:3]))
    h.close()
