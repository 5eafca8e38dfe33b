"""Randomised parity run of the tight-accuracy mode on the models with rows (round 5): Step 1 with G, Step 2 (rho > 0) and the beta-only objective (rho = 0), HIP library
against the numpy oracle's tight mode.  python tests/tools/tight_models_fuzz.py [seed] [members] -> JSON summary on the last line.  Members whose tight phase fell back to the
default point (info[10] = 4) are counted, not compared; members the oracle ends at a backed-off target are reported."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch  # noqa: F401
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 97001
members = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nxmax = int(sys.argv[3]) if len(sys.argv) > 3 else 8          # largest nx
rmax = int(sys.argv[4]) if len(sys.argv) > 4 else 3           # most rows of G and of C per stage
pmax = int(sys.argv[5]) if len(sys.argv) > 5 else 8
rng = np.random.default_rng(seed)
TOL = 2.0 ** -37
done = 0; worst = 0.0; fell_back = 0; skipped = 0; backed_off = 0; bad = []
t0 = time.time()
while done < members:
    p = int(rng.integers(1, pmax + 1)); nx = int(rng.integers(2, nxmax + 1)); mb = int(rng.integers(1, 5)); n = nx + mb
    ng = int(rng.integers(0, rmax + 1)); nc = int(rng.integers(0, rmax + 1)); model = ['G', 'step2', 'beta'][int(rng.integers(0, 3))]
    if model == 'G':
        nc = 0; ng = max(ng, 1)
    else:
        nc = max(nc, 1)
    nb = int(rng.integers(1, 4))
    A, B, H = co.gen_batch(int(rng.integers(1, 1 << 30)), nb, p, nx, mb)
    G = rng.standard_normal((nb, p, ng, n)) * float(rng.choice([0.3, 1.0, 3.0])); Cc = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32) if nc else np.zeros((nb, p), np.int32)
    for b in range(nb):
        for k in range(p):
            Cc[b, k, ncnt[b, k]:] = 0.0
    rho = 0.0 if model == 'beta' else float(rng.choice([1e-3, 1e-2, 1e-1]))
    h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb)
    h.set_tight(True, TOL)
    o = h.convexify_eq_batch(A, B, H, G) if model == 'G' else h.convexify_step2_batch(A, B, H, np.concatenate([G, Cc], axis=2), ncnt, rho)
    h.close()
    for b in range(nb):
        if o['info'][b, 13] != 0.0:
            skipped += 1; continue
        done += 1
        if int(o['info'][b, 10]) == 4:
            fell_back += 1; continue
        Cl = [Cc[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)] if nc else None
        kw = dict(G=G[b] if ng else None)
        if model != 'G':
            kw.update(C=Cl, rho=rho)
        r = co.sdp_step1(A[b], B[b], H[b], dict(tol=TOL, tight=True), **kw)
        if r['ipm_status'] != 'optimal' or r['mu_target'] != o['info'][b, 6]:
            backed_off += 1; continue
        Hc = H[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b] if ng else None, Fg=r.get('Fg'), C=Cl, F=r.get('F'))[0]
        e = float(np.linalg.norm(o['Hc'][b] - Hc) / np.linalg.norm(Hc))
        worst = max(worst, e)
        if not (e < 1e-8 and int(o['status'][b]) == 0):
            bad.append(dict(model=model, p=p, nx=nx, mb=mb, ng=ng, nc=nc, b=b, err=e, status=int(o['status'][b])))
print(json.dumps(dict(seed=seed, nx_max=nxmax, rows_max=rmax, p_max=pmax, members=done, worst_rel_err=worst, fell_back_to_default=fell_back, oracle_at_other_target=backed_off, already_convex=skipped, mismatches=bad,
                      seconds=round(time.time() - t0, 1))))
