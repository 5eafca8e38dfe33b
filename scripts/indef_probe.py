"""What the block factorisation does with a slightly indefinite diagonal block (frozen pivots): shifts counted, solution finite?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(2, 3, 1)
rng = np.random.default_rng(0)
for p, d in [(6, 48), (8, 300)]:
    blocks = [rng.standard_normal((d, d)) * 0.1 for _ in range(p)]
    T = np.zeros((p * d, p * d))
    Cc = np.zeros((p, d, d)); D = np.zeros((p, d, d))
    for k in range(p):
        kn = (k + 1) % p
        Cc[k] = blocks[k]
        T[k * d:(k + 1) * d, kn * d:(kn + 1) * d] += blocks[k]; T[kn * d:(kn + 1) * d, k * d:(k + 1) * d] += blocks[k].T
    T += np.eye(p * d) * (np.abs(np.linalg.eigvalsh(T)).max() + 1.0)
    for k in range(p):
        D[k] = T[k * d:(k + 1) * d, k * d:(k + 1) * d]
    rhs = rng.standard_normal((p, d))
    for eps in (0.0, 1e-13, 1e-9):
        D2 = D.copy()
        k = p // 2
        w, V = np.linalg.eigh(D2[k])
        # push the Schur complement of block k slightly negative along one direction: subtract (lambda_min + eps*max) v v' of the
        # exact pivot block would need the factor; use a crude stand-in: make D_k itself indefinite by -eps
        D2[k] = D2[k] - (w[0] + eps * w[-1]) * np.outer(V[:, 0], V[:, 0]) if eps else D2[k]
        x, ns = h.debug_block_solve(D2, Cc, rhs)
        print('p', p, 'd', d, 'eps', eps, 'nshift', ns, 'finite', np.isfinite(x).all(), 'max|x|', np.abs(x).max() if np.isfinite(x).all() else None)
