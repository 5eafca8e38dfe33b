"""sha256 of the raw outputs (Hc, P, kappa, status, iters) of a fixed problem set covering the five BASELINE shapes and the constrained models, for comparing two
builds of the library bit for bit:   python tests/tools/result_digest.py [path/to/other/libtunempc_hip.so]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
if len(sys.argv) > 1:
    alt = os.path.abspath(sys.argv[1])
    L.library_path = lambda: alt
from tunempc_amd import synthetic

print('library', L.library_path())


def dig(o, keys=('Hc', 'P', 'kappa', 'status', 'iters')):
    h = hashlib.sha256()
    for k in keys:
        h.update(np.ascontiguousarray(o[k]).tobytes())
    return h.hexdigest()[:16]


for name, (seed, nb, p, nx, mb) in {'c1 lqr-shaped': (1, 4, 1, 2, 1), 'c2 unicycle-shaped': (200000, 2, 30, 4, 1), 'c3 evaporation-shaped': (200100, 64, 50, 2, 2),
                                    'c4 bench shape': (100000, 8, 64, 24, 8), 'c5 long period': (300000, 1, 200, 24, 6)}.items():
    A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb)
    h = L.HipConvexifier(p, nx, mb, chunk=nb)
    print(f'{name:24s} {dig(h.convexify_batch(A, B, H))}', flush=True)
    h.close()
rng = np.random.default_rng(9)
p, nx, mb, ng, nc, nb = 6, 5, 2, 2, 3, 4
A, B, H = synthetic.gen_batch(77, nb, p, nx, mb)
J = rng.standard_normal((nb, p, ng + nc, nx + mb)); ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
for b in range(nb):
    for k in range(p):
        J[b, k, ng + ncnt[b, k]:] = 0.0
h = L.HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc)
print(f'{"Step 1 with G":24s} {dig(h.convexify_eq_batch(A, B, H, J[:, :, :ng]), ("Hc", "P", "Fg", "kappa", "status", "iters"))}')
print(f'{"Step 2":24s} {dig(h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2), ("Hc", "P", "FgF", "kappa", "status", "iters"))}')
h.close()
h = L.HipConvexifier(p, nx, mb, chunk=nb, step3=True)
print(f'{"Step 3":24s} {dig(h.convexify_step3_batch(A, B, H, 1e-2), ("Hc", "P", "T", "kappa", "status", "iters"))}')
h.close()
