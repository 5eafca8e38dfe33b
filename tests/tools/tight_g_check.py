"""Tight-accuracy mode with rows of G (round 5): HIP library against the numpy oracle's tight mode on small seeded shapes.
python tests/tools/tight_g_check.py  -> one line per shape (status, iterations, |Hc - oracle| / |oracle|, mu_target, kappa drop against the default)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch  # noqa: F401
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

rng = np.random.default_rng(5)
worst = 0.0
for seed, (p, nx, mb, ng, nb) in enumerate([(4, 3, 2, 2, 3), (6, 4, 2, 1, 3), (3, 5, 3, 3, 2), (8, 3, 1, 2, 2), (2, 6, 2, 2, 2), (1, 4, 2, 2, 2), (5, 4, 4, 3, 2), (6, 10, 4, 2, 2), (4, 16, 6, 3, 1)]):
    n = nx + mb
    A, B, H = co.gen_batch(100 + seed, nb, p, nx, mb)
    G = rng.standard_normal((nb, p, ng, n))
    h = HipConvexifier(p, nx, mb, ng=ng)
    o0 = h.convexify_eq_batch(A, B, H, G)
    h.set_tight(True, 2.0 ** -37)
    o = h.convexify_eq_batch(A, B, H, G)
    h.close()
    for b in range(nb):
        if o['info'][b, 13] != 0.0:
            print(f'p={p} nx={nx} mb={mb} b={b}: already convex (convexifier.py:83-85), nothing solved'); continue
        r = co.sdp_step1(A[b], B[b], H[b], dict(tol=2.0 ** -37, tight=True), G=G[b])
        Hc = H[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b], Fg=r['Fg'])[0]
        e = np.linalg.norm(o['Hc'][b] - Hc) / np.linalg.norm(Hc)
        ef = np.abs(o['Fg'][b] - r['Fg']).max() / np.abs(r['Fg']).max()
        worst = max(worst, e)
        print(f"p={p} nx={nx} mb={mb} ng={ng} b={b}: status {int(o['status'][b])} ipm {int(o['info'][b, 10])} iters {int(o['iters'][b])} (default {int(o0['iters'][b])}; oracle {r['iters']} {r['ipm_status']}) "
              f"Hc {e:.2e} Fg {ef:.2e} mu_t {o['info'][b, 6]:.2e} / {r['mu_target']:.2e} kappa drop {o0['kappa'][b] - o['kappa'][b]:.2e}", flush=True)
print(f'worst {worst:.2e}')
