"""Fuzz member (seed 88001, case 93, member 0: p = 9, nx = 7, mb = 4) that ends Feasible on the GPU and Optimal in the oracle: alone / in its batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier
A, B, H = synthetic.gen_batch(592540, 3, 9, 7, 4, sigP=0.6699498543473531)
for name, sel in (('member 0 alone', [0]),):
    h = HipConvexifier(9, 7, 4)
    o = h.convexify_batch(A[sel], B[sel], H[sel])
    tr = h.trace(len(sel))
    h.close()
    print(name, 'mu_t', o['info'][:, 6], 'kappa', o['kappa'], 'status', o['status'], 'iters', o['iters'], 'shifted pivots', o['info'][:, 11], 'min pivot ratio', o['info'][:, 15])
    b = sel.index(0)
    for r in tr[b]:
        if r[0] >= 9:
            print('   it %2d ph %.2f mu %.3e pinf %.1e ap %.3f raw %.3f stepn %.1e shifts %d' % (r[0], r[1], r[2], r[4], r[6], r[7], r[8], r[9]))
