import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from tunempc_amd._lib import HipConvexifier
import importlib.util
spec = importlib.util.spec_from_file_location('rsw', os.path.join(ROOT, 'scripts', 'robustness_sweep.py'))
src = open(os.path.join(ROOT, 'scripts', 'robustness_sweep.py')).read().split("rows = []")[0]
ns = {'__file__': os.path.join(ROOT, 'scripts', 'robustness_sweep.py')}; exec(src, ns)
p, nx, mb, sigP, cond_exp, rad = [float(x) if '.' in x else int(x) for x in sys.argv[1:7]]
nb = 8
ABH = [ns['gen'](7000 + 17 * b, p, nx, mb, sigP, cond_exp, rad) for b in range(nb)]
A = np.stack([x[0] for x in ABH]); B = np.stack([x[1] for x in ABH]); H = np.stack([x[2] for x in ABH])
h = HipConvexifier(p, nx, mb)
out = h.convexify_batch(A, B, H)
print('status', out['status'], 'iters', out['iters'], 'kappa', out['kappa'])
tr = h.trace(nb)
b = int(np.argmax(out['status'] != 0))
print('info', out['info'][b])
for row in tr[b]:
    if row[0] == 0: break
    print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
