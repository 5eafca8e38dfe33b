import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
from tunempc_amd import synthetic
np.set_printoptions(linewidth=250, precision=4)
p, nx, mb = 64, 24, 8
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
h = HipConvexifier(p, nx, mb, flags=FLAG_PROFILE)
t = time.time(); out = h.convexify_batch(A, B, H); print('time', time.time() - t)
print('status', np.bincount(out['status'], minlength=3), 'iters', out['iters'])
bad = np.where(out['status'] != 0)[0]
print('bad', bad)
for b in bad:
    print(b, 'kappa', out['kappa'][b], 'info', out['info'][b])
tr = h.trace(nb)
for b in list(bad)[:4] + [0]:
    print('--- trace of problem', b)
    for row in tr[b]:
        if row[0] == 0: break
        print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
print('profile', h.profile())
np.save(os.path.join(ROOT, 'gpurun_out', 'diag_iters.npy'), out['iters'])
