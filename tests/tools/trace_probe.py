"""Per-iteration trace (mu, infeasibilities, step lengths) of a few bench-shape problems."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A, B, H = synthetic.gen_batch(100000, nb, 64, 24, 8)
h = HipConvexifier(64, 24, 8, chunk=nb)
o = h.convexify_batch(A, B, H)
tr = h.trace(nb)
for b in range(min(nb, 3)):
    print('problem', b, 'iters', o['iters'][b], 'status', o['status'][b], 'kappa', o['kappa'][b])
    for r in tr[b]:
        if r[0] > 0:
            print('  it %2d ph %.2f mu %.2e tau %.3e pinf %.1e dinf %.1e ap %.3f ad/raw %.3f stepn %.1e' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
