"""Traces of the slowest members of the benchmark family (which rule keeps them iterating?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
p, nx, mb = 64, 24, 8
nb = 64
A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
h = HipConvexifier(p, nx, mb)
out = h.convexify_batch(A, B, H)
print('iters hist', np.bincount(out['iters']))
tr = h.trace(nb)
# phase structure: main iterations, centering iterations
for b in range(nb):
    rows = [r for r in tr[b] if r[0] > 0]
    nmain = sum(1 for r in rows if r[1] == 0); ncen = sum(1 for r in rows if r[1] == 1)
    print('b %2d iters %2d main %2d center %d  last steps %s' % (b, out['iters'][b], nmain, ncen, ' '.join('%.1e' % r[8] for r in rows[-5:])))
slow = np.where(out['iters'] == out['iters'].max())[0]
for b in slow[:3]:
    print('--- trace of problem', b, 'kappa', out['kappa'][b])
    for row in tr[b]:
        if row[0] == 0: break
        print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
