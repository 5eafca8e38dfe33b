"""Per-member detail of the smoke() batch: error, iterations and last centering steps on both sides."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier
p, nx, mb, nb = 6, 4, 2, 4
A, B, H = co.gen_batch(1234, nb, p, nx, mb)
h = HipConvexifier(p, nx, mb)
out = h.convexify_batch(A, B, H)
tr = h.trace(nb)
for b in range(nb):
    t = []
    r = co.sdp_step1(A[b], B[b], H[b], trace=t)
    rr = co.convexify_arrays(A[b], B[b], H[b])
    err = np.linalg.norm(out['Hc'][b] - rr['Hc']) / np.linalg.norm(rr['Hc'])
    rows = [x for x in tr[b] if x[0] > 0]
    print('b', b, 'err %.2e' % err, 'iters gpu/oracle', out['iters'][b], r['iters'], 'gpu last steps', ['%.1e' % x[8] for x in rows[-4:]], 'ipm', r['ipm_status'], out['info'][b, 10])
