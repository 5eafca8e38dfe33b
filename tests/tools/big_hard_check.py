"""Hard targets (cond(Hhat) = 1e5: the Schur complement turns numerically singular before the default target) on the generic per-stage kernels, n = 36 and 40:
status, back-offs, solver-independent invariants; with G rows as well.  python tests/tools/big_hard_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier


def calH(A, B, P):
    V = np.concatenate([A, B], axis=2); nx = A.shape[1]
    d = np.swapaxes(V, 1, 2) @ np.roll(P, -1, axis=0) @ V
    d[:, :nx, :nx] -= P
    return (d + np.swapaxes(d, 1, 2)) / 2


for (p, nx, mb, sigP, rad, ng) in [(6, 26, 10, 1.0, 0.5, 0), (4, 30, 10, 10.0, 0.5, 0), (5, 28, 8, 1.0, 0.5, 3)]:
    nb = 6
    probs = [synthetic.gen_problem(7100 + 17 * b, p, nx, mb, sigP=sigP, cond_exp=5, rad=rad) for b in range(nb)]
    A, B, H = (np.stack([q[i] for q in probs]) for i in range(3))
    h = HipConvexifier(p, nx, mb, ng=ng, chunk=nb)
    if ng:
        G = np.random.default_rng(1).standard_normal((nb, p, ng, nx + mb))
        out = h.convexify_eq_batch(A, B, H, G)
    else:
        out = h.convexify_batch(A, B, H)
    h.close()
    ks = []
    for b in range(nb):
        ev = np.linalg.eigvalsh(out['Hc'][b])
        mut0 = 2.0 ** np.round(np.log2(2.0 ** -25 * max(1.0, out['kappa'][b])))
        ks.append(int(np.round(np.log2(out['info'][b, 6] / mut0))))
        ok = ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-7)
        dH = calH(A[b], B[b], out['P'][b])
        if ng:
            dH = dH + np.einsum('ki,kij,kil->kjl', out['Fg'][b], G[b], G[b])
        sup = np.abs(out['Hc'][b] - H[b] - dH).max() / max(1.0, np.abs(H[b]).max())
        assert ok and sup < 1e-9, (b, ok, sup)
    print(f'n={nx + mb} p={p} ng={ng} cond 1e5: status {out["status"]} iterations {out["iters"]} back-offs {ks} shifted/lifted pivots {out["info"][:, 11].astype(int)}', flush=True)
