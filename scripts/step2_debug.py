import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
def run(nb, p, nx, mb, ng, nc, seed=100000):
    n = nx + mb
    A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(1)
    G = rng.standard_normal((nb, p, ng, n)); C = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            C[b, k, ncnt[b, k]:] = 0.0
    h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc)
    o1 = h.convexify_eq_batch(A, B, H, G) if ng else h.convexify_batch(A, B, H)
    print('shape', nb, p, nx, mb, ng, nc, 'G: iters', o1['iters'][:8], 'status', o1['status'][:8], 'kappa', o1['kappa'][:4])
    o2 = h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 1e-3)
    print('   step2: iters', o2['iters'][:8], 'status', o2['status'][:8], 'kappa', o2['kappa'][:4])
    tr = h.trace(nb)
    for row in tr[0][:6]:
        if row[0] == 0: break
        print('     it %2d ph %d mu %.3e tau %.6f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
    h.close()
run(2, 64, 24, 8, 2, 4)
run(2, 8, 24, 8, 2, 4)
run(2, 64, 4, 2, 2, 4)
run(2, 16, 12, 4, 2, 4)
run(2, 64, 24, 8, 0, 4)
