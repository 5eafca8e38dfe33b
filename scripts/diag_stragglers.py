import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
p, nx, mb = 64, 24, 8
A, B, H = synthetic.gen_batch(100000, 16, p, nx, mb)
h = HipConvexifier(p, nx, mb)
out = h.convexify_batch(A, B, H)
print('iters', out['iters'])
tr = h.trace(16)
for b in (3, 14, 4):
    print('--- trace of problem', b, 'kappa', out['kappa'][b])
    for row in tr[b]:
        if row[0] == 0: break
        print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
