"""A/B of a solver-rule change: the shipped library vs libtunempc_hip_alt.so on the same 64 problems."""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd import synthetic
p, nx, mb = 64, 24, 8
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
res = {}
import glob
libs = [('base', 'libtunempc_hip.so')] + [(os.path.basename(f)[len('libtunempc_hip_'):-3], os.path.basename(f)) for f in sorted(glob.glob(os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_alt*.so')))]
for tag, name in libs:
    import tunempc_amd._lib as L
    L = importlib.reload(L)
    L.library_path = lambda name=name: os.path.join(ROOT, 'tunempc_amd', 'lib', name)
    L._LIB = None
    h = L.HipConvexifier(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    res[tag] = out
    d = np.sqrt(((out['Hc'] - res['base']['Hc']) ** 2).sum(axis=(1, 2, 3)) / (res['base']['Hc'] ** 2).sum(axis=(1, 2, 3)))
    print(f"{tag:12s} status {np.bincount(out['status'], minlength=3)} iters max {out['iters'].max()} sum {out['iters'].sum()} hist {np.bincount(out['iters']).tolist()[12:]}  rel diff Hc vs base max {d.max():.2e}")
    if tag != 'base' and out['iters'].max() > res['base']['iters'].max():
        bw = int(np.argmax(out['iters']))
        print('   worst problem', bw, 'info', out['info'][bw][:12])
        for row in h.trace(nb)[bw]:
            if row[0] == 0: break
            print('     it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
    h.close()
