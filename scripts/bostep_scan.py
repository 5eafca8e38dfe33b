"""Where does the 'back mu_t off AND take the step' route of k_ctrl_b / k_ctrl_c fire?  Scan of hard-target configurations: per configuration the number of
centering iterations whose factorisation froze pivots, split into step taken (ap > 0: that route) and repeated (lift / plain back-off)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic

for (p, nx, mb, sigP, ce, rad) in [(30, 4, 1, 100.0, 5, 0.9), (30, 4, 1, 10.0, 6, 0.5), (30, 4, 1, 10.0, 8, 0.5), (8, 16, 4, 1.0, 7, 0.5), (8, 16, 4, 10.0, 8, 0.9), (5, 9, 6, 100.0, 7, 0.5),
                               (16, 12, 4, 10.0, 8, 0.9), (40, 9, 6, 10.0, 8, 0.5), (12, 6, 2, 100.0, 10, 0.9)]:
    nb = 16
    probs = [synthetic.gen_problem(7000 + 17 * b, p, nx, mb, sigP=sigP, cond_exp=ce, rad=rad) for b in range(nb)]
    A, B, H = (np.stack([q[i] for q in probs]) for i in range(3))
    h = HipConvexifier(p, nx, mb, chunk=nb)
    out = h.convexify_batch(A, B, H)
    tr = h.trace(nb)
    h.close()
    taken = retried = 0
    for b in range(nb):
        rows = tr[b][tr[b][:, 0] > 0]
        shifts = np.concatenate([[0.0], rows[:, 9]])
        for i, r in enumerate(rows):
            if int(r[1]) == 1 and shifts[i + 1] > shifts[i]:
                if r[6] > 0.0: taken += 1
                else: retried += 1
    print((p, nx, mb, sigP, ce, rad), 'optimal', int((out['status'] == 0).sum()), 'of', nb, 'max iters', int(out['iters'].max()), 'taken', taken, 'retried', retried, flush=True)
