"""Tight-accuracy mode on a large-block shape (plain model, p = 16, nx = 40, n = 48, batch 32): time of the default solve and of the tight solve."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier

nb, p, nx, mb = 32, 16, 40, 8
A, B, H = synthetic.gen_batch(200200, nb, p, nx, mb)
h = HipConvexifier(p, nx, mb, chunk=nb)
h.convexify_batch(A, B, H)
t0 = time.perf_counter(); d = h.convexify_batch(A, B, H); t1 = time.perf_counter()
h.set_tight(True)
h.convexify_batch(A, B, H)
t2 = time.perf_counter(); t = h.convexify_batch(A, B, H); t3 = time.perf_counter()
h.close()
N = 2 * p * (nx + mb) + 1
print(f'default: {1e3 * (t1 - t0):.0f} ms ({nb * p / (t1 - t0):.0f} stage-conv/s), {d["iters"].mean():.1f} iterations, gap N mu_t / kappa = {N * (d["info"][:, 6] / d["kappa"]).max():.2e}')
print(f'tight  : {1e3 * (t3 - t2):.0f} ms ({nb * p / (t3 - t2):.0f} stage-conv/s), {t["iters"].mean():.1f} iterations, gap N mu_t / kappa = {N * (t["info"][:, 6] / t["kappa"]).max():.2e}, '
      f'optimal {(t["status"] == 0).sum()} / {nb}, members at the tight target {(t["info"][:, 6] < 1e-3 * d["info"][:, 6]).sum()}, kappa drop mean {(d["kappa"] - t["kappa"]).mean():.2e}')
