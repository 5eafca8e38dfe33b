"""Step 1 with the equality-constraint term G on the GPU (tmpc_convexify_eq_batch_host) against the structured oracle.
Usage: python tests/tools/eq_check.py   (needs a GPU; prints relative errors and iteration counts per case)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

CASES = [(20, 2, 3, 3, 2, 2), (0, 3, 3, 3, 2, 1), (30, 2, 2, 3, 1, 2), (20, 2, 1, 3, 1, 1), (7, 2, 5, 4, 2, 3), (11, 3, 4, 6, 3, 4),
         (5, 2, 8, 8, 4, 2)]
worst = 0.0
for seed, nb, p, nx, mb, ng in CASES:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(1000 + seed)
    G = rng.standard_normal((nb, p, ng, nx + mb))
    cv = HipConvexifier(p, nx, mb, chunk=nb, ng=ng)
    out = cv.convexify_eq_batch(A, B, H, G)
    base = cv.convexify_batch(A, B, H)
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b], G=G[b])
        r0 = co.convexify_arrays(A[b], B[b], H[b])
        eH = np.linalg.norm(out['Hc'][b] - r['Hc']) / np.linalg.norm(r['Hc'])
        r.setdefault('Fg', np.zeros((p, ng)))      # early exit (already convex): no multipliers
        eF = np.linalg.norm(out['Fg'][b] - r['Fg']) / max(1e-300, np.linalg.norm(r['Fg']))
        e0 = np.linalg.norm(base['Hc'][b] - r0['Hc']) / np.linalg.norm(r0['Hc'])
        worst = max(worst, eH, e0)
        print(f"seed {seed} b{b} p{p} nx{nx} mb{mb} ng{ng}: Hc {eH:.2e} Fg {eF:.2e} (|Fg| {np.linalg.norm(r['Fg']):.2e}) kappa {out['kappa'][b]:.6g}/{r['kappa']:.6g}"
              f" it {out['iters'][b]}/{r['iters']} st {out['status'][b]}/{r['status']}  noG: {e0:.2e} kappa0 {r0['kappa']:.6g}")
print('worst', worst)
