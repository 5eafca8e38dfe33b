"""Side-by-side traces of the Step 3 model: HIP path vs the structured oracle (one small problem)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier
seed, p, nx, mb, rho = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (300, 2, 2, 1, 1e-3)
A, B, H = co.gen_batch(seed, 1, p, nx, mb)
tr = []
r = co.sdp_step1(A[0], B[0], H[0], rho=rho, force=True, trace=tr)
print('oracle: iters', r['iters'], r['ipm_status'], 'kappa', r['kappa'], 'objective', r.get('objective'))
for t in tr:
    print('  it %2d ph%d mu %.3e tau %.8f pinf %.2e dinf %.2e' % (t['it'], t['phase'], t['mu'], t['tau'], t['pinf'], t['dinf']))
h = HipConvexifier(p, nx, mb, step3=True, chunk=1)
o = h.convexify_step3_batch(A, B, H, rho)
print('gpu: status', o['status'], 'iters', o['iters'], 'kappa', o['kappa'], 'info', o['info'][0])
for row in h.trace(1)[0]:
    if row[0] > 0:
        print('  it %2d ph%.2f mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3e stepn %.2e shifts %d' % (row[0] - 1, row[1], row[2], row[3], row[4], row[5], row[6], row[7], row[8], row[9]))
st, dHc = co.check_convergence(A[0], B[0], H[0], r['P'], r['ipm_status'], T=r['T'])[:2]
print('Hc rel diff', np.linalg.norm(o['Hc'][0] - H[0] - dHc) / np.linalg.norm(H[0] + dHc), 'T rel diff', np.linalg.norm(o['T'][0] - r['T']) / np.linalg.norm(r['T']))
