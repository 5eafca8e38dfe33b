"""Hard targets (cond(Hhat) = 1e5: the Schur matrix turns numerically singular before the default mu_t): HIP path against the oracle.
Both back mu_t off by powers of two until the centering converges; they need not need the same number of back-offs (different pivoting
safeguards: frozen pivots + a diagonal lift on the GPU, a uniform relative shift in the oracle), so the comparison is made member by member
where the final mu_t agree, and the achieved mu_t are printed for all.
    python tests/tools/hard_target_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch  # noqa: F401,E402
from tunempc_amd import synthetic  # noqa: E402
from tunempc_amd._lib import HipConvexifier  # noqa: E402
import convexify_oracle as co  # noqa: E402

rows = []
for (p, nx, mb, sigP, cond_exp, rad) in [(30, 4, 1, 10.0, 5, 0.5), (30, 4, 1, 100.0, 5, 0.9), (8, 16, 4, 1.0, 5, 0.5), (5, 9, 6, 100.0, 5, 0.5)]:
    nb = 8
    probs = [synthetic.gen_problem(7000 + 17 * b, p, nx, mb, sigP=sigP, cond_exp=cond_exp, rad=rad) for b in range(nb)]
    A, B, H = (np.stack([q[i] for q in probs]) for i in range(3))
    h = HipConvexifier(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    h.close()
    for b in range(nb):
        r = co.sdp_step1(A[b], B[b], H[b])
        st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'])[:2]
        Hc = H[b] + dHc
        mut_g = out['info'][b, 6]
        kg = np.log2(mut_g / 2.0 ** np.round(np.log2(2.0 ** -25 * max(1.0, out['kappa'][b]))))
        ko = np.log2(r['mu_target'] / 2.0 ** np.round(np.log2(2.0 ** -25 * max(1.0, r['kappa']))))
        err = np.linalg.norm(out['Hc'][b] - Hc) / np.linalg.norm(Hc)
        rows.append((p, nx, mb, sigP, rad, b, int(out['status'][b]), st, int(out['iters'][b]), r['iters'], kg, ko, err, abs(out['kappa'][b] - r['kappa']) / r['kappa']))
        print('p %2d nx %2d sigP %5.1f rad %.1f b %d | status gpu %d oracle %d | iters %2d %2d | back-offs gpu %.0f oracle %.0f | Hc rel %.2e kappa rel %.2e'
              % (p, nx, sigP, rad, b, out['status'][b], st, out['iters'][b], r['iters'], kg, ko, err, rows[-1][-1]), flush=True)
same = [r for r in rows if r[10] == r[11]]
print('members', len(rows), 'optimal on the GPU', sum(r[6] == 0 for r in rows), 'in the oracle', sum(r[7] == 0 for r in rows))
print('same final mu_t:', len(same), 'worst Hc', max(r[12] for r in same), '; without back-off:', max([r[12] for r in same if r[10] == 0] or [0]),
      '; with:', max([r[12] for r in same if r[10] > 0] or [0]))
print('worst kappa difference over all members (relative):', max(r[13] for r in rows))
