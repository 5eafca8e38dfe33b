"""Replays one case of parity_fuzz.py (same random stream) and prints the GPU traces and the oracle's of one member/model.
Usage: python tests/tools/fuzz_case.py <case> <member> <plain|G|step2|beta|step3> [seed] [pmax] [nxmax]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier
want, bsel, model = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 2024)
pmax = int(sys.argv[5]) if len(sys.argv) > 5 else 12
nxmax = int(sys.argv[6]) if len(sys.argv) > 6 else 8
for case in range(want + 1):
    p = int(rng.integers(1, pmax + 1)); nx = int(rng.integers(1, nxmax + 1)); mb = int(rng.integers(1, 5))
    n = nx + mb
    ng = int(rng.integers(1, 4)); nc = int(rng.integers(1, 5))
    seed = int(rng.integers(0, 10 ** 6))
    nb = 3
    sig = float(10.0 ** rng.uniform(-0.5, 1.0))
    gs = float(10.0 ** rng.uniform(-1, 1))
    G = gs * rng.standard_normal((nb, p, ng, n)); C = gs * rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    rho = float(10.0 ** rng.uniform(-3, 0))
A, B, H = co.gen_batch(seed, nb, p, nx, mb, sigP=sig)
for b in range(nb):
    for k in range(p):
        C[b, k, ncnt[b, k]:] = 0.0
print('case', want, 'p', p, 'nx', nx, 'mb', mb, 'ng', ng, 'nc', nc, 'seed', seed, 'sigP', sig, 'gs', gs, 'rho', rho)
h = HipConvexifier(p, nx, mb, ng=ng, nc=nc)
if model == 'plain':
    o = h.convexify_batch(A, B, H)
elif model == 'G':
    o = h.convexify_eq_batch(A, B, H, G)
elif model == 'step3':
    h.close()
    h = HipConvexifier(p, nx, mb, step3=True)
    o = h.convexify_step3_batch(A, B, H, rho)
else:
    o = h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 0.0 if model == 'beta' else rho)
tr = h.trace(nb)
b = bsel
print('gpu status', o['status'][b], 'iters', o['iters'][b], 'kappa', o['kappa'][b], 'info', o['info'][b, 10:16])
for row in tr[b]:
    if row[0] == 0: break
    print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
t = []
Cl = [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
if model == 'step3':
    r = co.sdp_step1(A[b], B[b], H[b], trace=t, rho=rho, force=True)
else:
    r = co.sdp_step1(A[b], B[b], H[b], trace=t, G=None if model == 'plain' else G[b], C=Cl if model in ('step2', 'beta') else None,
                     rho=rho if model in ('step2', 'beta') else None, cost_free=(model == 'beta'))
print('oracle', r['ipm_status'], 'iters', r['iters'], 'kappa', r['kappa'], 'shift', r['shift'])
if model == 'plain':
    Hco = H[b] + co.convex_hessian_suppl(A[b], B[b], r['P'])[0]
    print('rel. Frobenius error of Hc, GPU vs oracle: %.3e' % (np.linalg.norm(o['Hc'][b] - Hco) / np.linalg.norm(Hco)))
for x in t:
    print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e' % (x['it'] + 1, x['phase'], x['mu'], x['tau'], x['pinf'], x['dinf']))
