"""How well is Hc determined by the SDP?  (CPU only, oracle.)

The SDP of convexifier.py:213-308 minimises beta alone.  Its minimiser is a FACE: stages whose condition number stays below kappa* leave P free.
An interior-point solver returns a point near the central path at the barrier parameter mu where it stops; the path ends (mu -> 0) at the
analytic centre of the optimal face.  This script measures, on the oracle:
  (a) the identity family (Hhat = I: the answer Hc = I is unique): distance of the returned point to I, and kappa - 1;
  (b) for generic members: how far Hc at the returned centred point (mu_t = 2^-25 kappa) is from the path's limit, using the Taylor model of the path
      (orders 1..K from ONE factorisation) -- and how much Hc moves when the stopping mu changes by 4x and 16x, i.e. the spread two correct solvers
      (or one solver at two tolerances) show on Hc although both are optimal to their tolerance;
  (c) the same for kappa.
    python tests/tools/path_sensitivity.py > profiles/r3_path_sensitivity.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402


def hc_of(A, B, H, s, y):
    tau, alpha, Pbar = y
    return H + co.symmetrize(co.calH(A, B, Pbar / (s * alpha)))


def run(A, B, H, tol, K):
    r = co.sdp_step1(A, B, H, dict(tol=tol, extrap=K))
    y0 = (r['kappa'], r['alpha'], r['Pbar'])
    ys = [y0]
    acc = y0
    for tm in r['extrap_terms']:
        acc = (acc[0] + tm[0], acc[1] + tm[1], acc[2] + tm[2])
        ys.append(acc)
    return r, [hc_of(A, B, H, r['s'], y) for y in ys], [y[0] for y in ys]


rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
K = 6
print('(a) identity family (Hhat = I  =>  Hc = I, kappa* = 1 whatever the solver)')
for seed, p, nx, mb in [(77, 5, 4, 2), (78, 16, 12, 4)]:
    A, B, H, _, _ = co.gen_problem(seed, p, nx, mb, identity=True)
    r, hcs, taus = run(A, B, H, 2.0 ** -25, 2)
    n = nx + mb
    print(f'  p={p:3d} n={n:2d}: returned point max|Hc - I| = {np.abs(hcs[0] - np.eye(n)).max():.2e}, kappa - 1 = {taus[0] - 1:.3e} (= N mu_t = {(2 * p * n + 1) * r["mu_target"]:.3e});'
          f' first-order model of the limit: max|Hc - I| = {np.abs(hcs[1] - np.eye(n)).max():.1e}, kappa - 1 = {taus[1] - 1:.1e}')
print()
print(f'(b, c) generic members: the returned centred point against the limit point of the central path (Taylor model, orders 1..{K}; the order-{K} value is the reference)')
print('        "spread": Hc at the returned point for stopping parameters 4x and 16x larger, relative to the default one')
for seed, p, nx, mb in [(10, 4, 3, 2), (11, 8, 4, 1), (5, 16, 4, 1), (3, 12, 6, 3), (100000, 16, 12, 4), (100001, 16, 12, 4)]:
    A, B, H, _, _ = co.gen_problem(seed, p, nx, mb)
    r, hcs, taus = run(A, B, H, 2.0 ** -25, K)
    r4, hcs4, taus4 = run(A, B, H, 2.0 ** -23, 1)
    r16, hcs16, taus16 = run(A, B, H, 2.0 ** -21, 1)
    lim, tl = hcs[-1], taus[-1]
    print(f'  seed {seed:6d} p={p:3d} n={nx + mb:2d} kappa={taus[0]:.6f}: |Hc(mu_t) - limit| / |limit| = {rel(hcs[0], lim):.1e}'
          f'  (orders 1..{K - 1}: ' + ' '.join(f'{rel(h, lim):.0e}' for h in hcs[1:-1]) + ')')
    print(f'        spread of Hc: mu_t x4: {rel(hcs4[0], hcs[0]):.1e}, x16: {rel(hcs16[0], hcs[0]):.1e};'
          f'   kappa(mu_t) - limit = {taus[0] - tl:.1e} (relative {abs(taus[0] - tl) / tl:.1e}; orders 1..3: ' + ' '.join(f'{abs(t - tl) / tl:.0e}' for t in taus[1:4]) + ')')
