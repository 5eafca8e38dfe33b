"""Smallest relative Cholesky pivot of the Schur factorisations (info[15]) for the benchmark family and a hard family."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
src = open(os.path.join(ROOT, 'scripts', 'robustness_sweep.py')).read().split("rows = []")[0]
ns = {'__file__': os.path.join(ROOT, 'scripts', 'robustness_sweep.py')}; exec(src, ns)
A, B, H = synthetic.gen_batch(100000, 32, 64, 24, 8)
h = HipConvexifier(64, 24, 8)
out = h.convexify_batch(A, B, H)
mp = out['info'][:, 15]
print('benchmark family: min relative pivot over the run: min %.2e median %.2e max %.2e; status %s' % (mp.min(), np.median(mp), mp.max(), np.bincount(out['status'], minlength=3)))
h.close()
for (p, nx, mb, sigP, ce, rad) in [(8, 16, 4, 1.0, 3, 0.5), (8, 16, 4, 1.0, 5, 0.5), (5, 9, 6, 100.0, 5, 0.5), (30, 4, 1, 1.0, 5, 0.9), (5, 9, 6, 1.0, 1, 0.9)]:
    ABH = [ns['gen'](7000 + 17 * b, p, nx, mb, sigP, ce, rad) for b in range(8)]
    A = np.stack([x[0] for x in ABH]); B = np.stack([x[1] for x in ABH]); H = np.stack([x[2] for x in ABH])
    h = HipConvexifier(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    print((p, nx, mb, sigP, ce, rad), 'status', out['status'].tolist(), 'minpiv', ['%.1e' % v for v in out['info'][:, 15]], 'kappa', ['%.0f' % v for v in out['kappa']])
    h.close()
