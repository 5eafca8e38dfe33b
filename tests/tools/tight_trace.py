import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import convexify_oracle as co
from tunempc_amd import _lib
np.set_printoptions(linewidth=250, precision=3)
seed, nb, p, nx, mb = [int(x) for x in sys.argv[1:6]]
A, B, H = co.gen_batch(seed, nb, p, nx, mb)
h = _lib.HipConvexifier(p, nx, mb, chunk=nb)
h.set_tight(True, 2.0 ** -37)
r = h.convexify_batch(A, B, H)
tr = h.trace(nb)
print('status', r['status'], 'iters', r['iters'], 'info10(ipm)', r['info'][:, 10], 'shifts', r['info'][:, 11])
for b in range(nb):
    print('problem', b)
    for row in tr[b]:
        if row[0] > 0: print('  it %2d ph %.2f mu %.3e tau %.10f pinf %.1e dinf %.1e ap %.3g ad/raw %.3g stepn %.2e shifts %d' % tuple(row))
