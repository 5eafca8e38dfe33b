import sys, time, numpy as np
sys.path.insert(0, '.')
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
for (nb, p, nx, mb) in [(1, 30, 4, 1), (256, 50, 2, 2), (64, 40, 9, 6)]:
    A, B, H = synthetic.gen_batch(2000, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb)
    h.convexify_batch(A, B, H)
    t = time.perf_counter(); out = h.convexify_batch(A, B, H); el = time.perf_counter() - t
    print((nb, p, nx, mb), 'host-to-host %.2f ms, iters %d -> %.3f ms per iteration' % (el * 1e3, out['iters'].max(), el * 1e3 / out['iters'].max()))
    h.close()
