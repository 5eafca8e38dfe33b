"""Step 2 (active-constraint multipliers + norm terms) on the GPU against the structured oracle.
Usage: python tests/tools/step2_check.py   (needs a GPU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

CASES = [(20, 3, 3, 2, 0, [2, 0, 1], 1e-3), (0, 3, 3, 2, 2, [1, 2, 0], 1.0), (30, 2, 3, 1, 1, [1, 1], 1e-3), (20, 1, 3, 1, 0, [2], 1.0),
         (7, 5, 4, 2, 3, [0, 3, 1, 2, 3], 1e-2), (11, 4, 6, 3, 2, [4, 0, 8, 1], 1e-3)]
worst = 0.0
for seed, p, nx, mb, ng, ncs, rho in CASES:
    nb = 2
    n = nx + mb
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 5)
    nc = max(ncs)
    J = np.zeros((nb, p, ng + nc, n)); ncnt = np.zeros((nb, p), np.int32)
    Gs = []; Cs = []
    for b in range(nb):
        G = rng.standard_normal((p, ng, n)) if ng else None
        C = [rng.standard_normal((c, n)) if c else None for c in ncs]
        Gs.append(G); Cs.append(C)
        for k in range(p):
            if ng: J[b, k, :ng] = G[k]
            if ncs[k]: J[b, k, ng:ng + ncs[k]] = C[k]
            ncnt[b, k] = ncs[k]
    cv = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc)
    out = cv.convexify_step2_batch(A, B, H, J, ncnt, rho)
    for b in range(nb):
        if np.linalg.eigvalsh(H[b])[:, 0].min() > 0:
            print('seed', seed, b, 'already convex: iters', out['iters'][b]); continue
        r = co.sdp_step1(A[b], B[b], H[b], G=Gs[b], C=Cs[b], rho=rho)
        st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], G=Gs[b], Fg=r.get('Fg'), C=Cs[b], F=r['F'])[:2]
        eH = np.linalg.norm(out['dHc'][b] - dHc) / np.linalg.norm(H[b] + dHc)
        Fo = np.zeros((p, ng + nc))
        for k in range(p):
            if ng: Fo[k, :ng] = r['Fg'][k]
            if ncs[k]: Fo[k, ng:ng + ncs[k]] = r['F'][k]
        eF = np.linalg.norm(out['FgF'][b] - Fo) / max(1e-300, np.linalg.norm(Fo))
        worst = max(worst, eH)
        print(f"seed {seed} b{b} p{p} nx{nx} mb{mb} ng{ng} nc{ncs} rho {rho}: Hc {eH:.2e} F {eF:.2e} kappa {out['kappa'][b]:.8g}/{r['kappa']:.8g}"
              f" it {out['iters'][b]}/{r['iters']} st {out['status'][b]}/{st}")
print('worst', worst)
