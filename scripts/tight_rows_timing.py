"""One default and one tight call of the Step 2 model (rows of G, ragged rows of C, norm terms) at the bench stage shape, host-buffer entry, for timing / rocprofv3
--kernel-trace --stats.   python scripts/tight_rows_timing.py [batch, default 256] [ng, default 2] [nc, default 3]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier  # noqa: E402
from tunempc_amd import synthetic  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p, nx, mb, rho = 64, 24, 8, 1e-2
n = nx + mb
A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
rng = np.random.default_rng(5)
J = rng.standard_normal((nb, p, ng + nc, n)); cnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
for b in range(nb):
    for k in range(p):
        J[b, k, ng + cnt[b, k]:] = 0.0
h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb)
h.convexify_step2_batch(A[:4], B[:4], H[:4], J[:4], cnt[:4], rho)
t0 = time.perf_counter(); o0 = h.convexify_step2_batch(A, B, H, J, cnt, rho); t1 = time.perf_counter()
h.set_tight(True)
h.convexify_step2_batch(A[:4], B[:4], H[:4], J[:4], cnt[:4], rho)
t2 = time.perf_counter(); o1 = h.convexify_step2_batch(A, B, H, J, cnt, rho); t3 = time.perf_counter()
h.close()
print(f'Step 2 model, p = {p}, n = {n}, {ng} + <= {nc} rows, batch {nb}: default {t1 - t0:.3f} s ({nb * p / (t1 - t0):.0f} stage-conv/s, {o0["iters"].mean():.1f} iterations, '
      f'{int((o0["status"] == 0).sum())}/{nb} Optimal); tight 2^-37: {t3 - t2:.3f} s ({nb * p / (t3 - t2):.0f} stage-conv/s, {o1["iters"].mean():.1f} iterations, '
      f'{int((o1["status"] == 0).sum())}/{nb} Optimal, {int((o1["info"][:, 10] == 4).sum())} fell back), kappa drop mean {np.mean(o0["kappa"] - o1["kappa"]):.3e}')
