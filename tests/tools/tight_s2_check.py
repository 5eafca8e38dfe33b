"""Tight-accuracy mode on the Step 2 model (round 5): HIP library against the numpy oracle's tight mode on small seeded shapes.
python tests/tools/tight_s2_check.py -> one line per member."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch  # noqa: F401
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

rng = np.random.default_rng(7)
worst = 0.0
rho = 1e-2
for seed, (p, nx, mb, ng, nc, nb) in enumerate([(4, 3, 2, 1, 2, 3), (5, 4, 2, 0, 2, 3), (3, 5, 3, 2, 3, 2), (6, 3, 1, 1, 1, 2), (2, 6, 2, 2, 2, 2), (1, 4, 2, 1, 2, 2), (6, 10, 4, 2, 3, 2), (4, 16, 6, 1, 4, 1)]):
    n = nx + mb
    A, B, H = co.gen_batch(300 + seed, nb, p, nx, mb)
    G = rng.standard_normal((nb, p, ng, n)); Cc = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            Cc[b, k, ncnt[b, k]:] = 0.0
    J = np.concatenate([G, Cc], axis=2)
    h = HipConvexifier(p, nx, mb, ng=ng, nc=nc)
    o0 = h.convexify_step2_batch(A, B, H, J, ncnt, rho)
    h.set_tight(True, 2.0 ** -37)
    o = h.convexify_step2_batch(A, B, H, J, ncnt, rho)
    h.close()
    for b in range(nb):
        if o['info'][b, 13] != 0.0:
            print(f'p={p} nx={nx} mb={mb} b={b}: already convex (convexifier.py:83-85), nothing solved'); continue
        Cl = [Cc[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        r = co.sdp_step1(A[b], B[b], H[b], dict(tol=2.0 ** -37, tight=True), G=G[b] if ng else None, C=Cl, rho=rho)
        Hc = H[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b] if ng else None, Fg=r.get('Fg'), C=Cl, F=r['F'])[0]
        e = np.linalg.norm(o['Hc'][b] - Hc) / np.linalg.norm(Hc)
        worst = max(worst, e)
        print(f"p={p} nx={nx} mb={mb} ng={ng} nc={nc} b={b}: status {int(o['status'][b])} ipm {int(o['info'][b, 10])} iters {int(o['iters'][b])} (default {int(o0['iters'][b])}; oracle {r['iters']} {r['ipm_status']}) "
              f"Hc {e:.2e} mu_t {o['info'][b, 6]:.2e} / {r['mu_target']:.2e} kappa {o['kappa'][b]:.12f} / {r['kappa']:.12f}", flush=True)
print(f'worst {worst:.2e}')
