"""Sweep of Step 1 with G and of the Step 2 model over shapes, rho and the scaling of the Jacobian rows: status counts and the
solver-independent invariants (Hc > 0, cond <= kappa, supplement = calH(P) + G'FgG + C'FC, multipliers >= 0)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

rows = []
for (p, nx, mb, ng, nc) in [(1, 3, 1, 1, 2), (2, 4, 2, 2, 3), (5, 9, 6, 3, 4), (30, 4, 1, 1, 2), (8, 16, 4, 2, 8), (40, 9, 6, 3, 3), (64, 24, 8, 2, 4)]:
    n = nx + mb
    nb = 8
    h = HipConvexifier(p, nx, mb, ng=ng, nc=nc)
    A, B, H = co.gen_batch(4000 + p, nb, p, nx, mb)
    for gs in (0.1, 1.0, 10.0):
        rng = np.random.default_rng(p * 7 + int(gs * 10))
        G = gs * rng.standard_normal((nb, p, ng, n)); C = gs * rng.standard_normal((nb, p, nc, n))
        ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
        for b in range(nb):
            for k in range(p):
                C[b, k, ncnt[b, k]:] = 0.0
        runs = [('G', None, h.convexify_eq_batch(A, B, H, G))]
        for rho in (1e-3, 1.0):
            runs.append(('step2', rho, h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, rho)))
        for tag, rho, out in runs:
            ok = 0
            for b in range(nb):
                Fo = out['Fg'][b] if tag == 'G' else out['FgF'][b]
                ev = np.linalg.eigvalsh(out['Hc'][b])
                early = bool(out['info'][b, 13])
                Cl = None if tag == 'G' else [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
                Fl = None if tag == 'G' else [Fo[k, ng:ng + ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
                ref = co.convex_hessian_suppl(A[b], B[b], out['P'][b], G=G[b], Fg=Fo[:, :ng], C=Cl, F=Fl)[0]
                struct = np.abs(out['Hc'][b] - H[b] - (0 if early else ref)).max() / max(1.0, np.abs(H[b]).max())
                good = out['status'][b] == 0 and ev.min() > 0 and struct < 1e-9 and (Fo >= 0).all() and \
                    (early or (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-7))
                ok += bool(good)
            rows.append(dict(p=p, nx=nx, mb=mb, ng=ng, nc=nc, gscale=gs, model=tag, rho=rho, ok=ok, nb=nb, iters_max=int(out['iters'].max()),
                             status=np.bincount(out['status'], minlength=3).tolist()))
            if ok != nb:
                print('NOT ALL OK', rows[-1])
    h.close()
print('cases', len(rows), 'members ok', sum(r['ok'] for r in rows), 'of', sum(r['nb'] for r in rows), 'max iterations', max(r['iters_max'] for r in rows))
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'step2_sweep.json'), 'w'))
