"""Hard targets at the bench shape (p = 64, nx = 24, m = 8): 16 members at cond(Hhat) = 1e3 / 1e5, statuses, back-offs, invariants, time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier
p, nx, mb, nb = 64, 24, 8, 16
import sys as _s
h = HipConvexifier(p, nx, mb, chunk=nb)
if len(_s.argv) > 1:
    h.set_tuning(lowp_switch=float(_s.argv[1]))      # (round 6: the single-precision updates off / at another switch)
    print('lowp_switch', float(_s.argv[1]))
for cond_exp in (1, 3, 5):
    probs = [synthetic.gen_problem(4000 + 7 * b, p, nx, mb, sigP=10.0, cond_exp=cond_exp, rad=0.9) for b in range(nb)]
    A, B, H = (np.stack([q[i] for q in probs]) for i in range(3))
    t = time.perf_counter(); o = h.convexify_batch(A, B, H); dt = time.perf_counter() - t
    ev = np.linalg.eigvalsh(o['Hc'])
    mut0 = 2.0 ** np.round(np.log2(2.0 ** -25 * np.maximum(1.0, o['kappa'])))
    back = np.round(np.log2(o['info'][:, 6] / mut0)).astype(int)
    print('cond 1e%d: status %s iters %s back-offs %s  min eig Hc > 0: %s  cond <= kappa: %s  %.2f s' % (
        cond_exp, np.bincount(o['status'], minlength=3).tolist(), o['iters'].tolist(), back.tolist(), bool(ev.min() > 0),
        bool(((ev[:, :, -1] / ev[:, :, 0]).max(axis=1) <= o['kappa'] * (1 + 1e-7)).all()), dt), flush=True)
    import hashlib; print('   kappa', np.array2string(o['kappa'][:4], precision=12), 'digest', hashlib.sha256(o['Hc'].tobytes()).hexdigest()[:12], flush=True)
h.close()
