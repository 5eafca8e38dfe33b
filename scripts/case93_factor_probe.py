"""The Schur matrix of the fuzz member that ended Feasible (tests/tools/case93_schur.npz, from the oracle's iterate): block solve on the GPU, plain and with the lifted diagonal."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from tunempc_amd._lib import HipConvexifier
z = np.load(os.path.join(ROOT, 'tests', 'tools', 'case93_schur.npz')); D, C = z['D'], z['C']
p, d, _ = D.shape
h = HipConvexifier(p, 7, 4)
rng = np.random.default_rng(0)
rhs = rng.standard_normal((p, d))
def dense(Dm):
    T = np.zeros((p * d, p * d))
    for k in range(p):
        T[k*d:(k+1)*d, k*d:(k+1)*d] += Dm[k]; kn = (k + 1) % p
        T[k*d:(k+1)*d, kn*d:(kn+1)*d] += C[k]; T[kn*d:(kn+1)*d, k*d:(k+1)*d] += C[k].T
    return T
for lift in (0.0, 1e-13, 1e-12, 1e-11, 1e-10, 1e-8):
    Dm = D.copy()
    for k in range(p):
        Dm[k][np.arange(d), np.arange(d)] *= (1.0 + lift)
    x, nshift = h.debug_block_solve(Dm, C, rhs)
    T = dense(Dm)
    xr = np.linalg.solve(T, rhs.ravel()).reshape(p, d)
    res = np.linalg.norm(T @ x.ravel() - rhs.ravel()) / np.linalg.norm(rhs)
    print('lift %.0e: frozen pivots %d, finite %s, |x - x_dense|/|x_dense| %.2e, residual %.2e' % (lift, nshift, np.isfinite(x).all(), np.linalg.norm(x - xr) / np.linalg.norm(xr), res))
h.close()
