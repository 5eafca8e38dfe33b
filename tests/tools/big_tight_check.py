"""Tight-accuracy mode at 32 < n <= 64 (dd stage matrices in global scratch, k_dd_schur reading its factors from global memory at nx > 35) against the C++ port's tight mode."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
import cpu_ipm
from tunempc_amd._lib import HipConvexifier

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for (seed, nb, p, nx, mb, lt) in [(501, 2, 3, 24, 10, 37), (502, 1, 2, 30, 8, 37), (503, 1, 2, 40, 8, 37), (504, 2, 4, 20, 16, 39)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=nb)
    d0 = h.convexify_batch(A, B, H)
    h.set_tight(True, 2.0 ** -lt)
    t0 = time.time()
    o = h.convexify_batch(A, B, H)
    t1 = time.time()
    h.close()
    c = cpu_ipm.convexify_batch(A, B, H, tol=2.0 ** -lt, threads=16, tight=True)
    print(f'n={nx + mb} nx={nx} p={p} 2^-{lt}: HIP status {o["status"]} iters {o["iters"]} mu_t {o["info"][:, 6]} ({t1 - t0:.1f}s) cpu {c["status"]} {c["iters"]} mu_t {c["mu_t"]} ({time.time() - t1:.1f}s) '
          f'Hc err {max(rel(o["Hc"][b], c["Hc"][b]) for b in range(nb)):.2e} kappa diff {np.abs(o["kappa"] - c["kappa"]).max():.2e} default-vs-tight Hc {rel(d0["Hc"], o["Hc"]):.2e}', flush=True)
