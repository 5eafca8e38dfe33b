"""Step 3 on the generic per-stage kernels: (a) at n <= 32 against the tuned kernels (debug flag 64), (b) at n = 34 against the numpy oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)

for (seed, nb, p, nx, mb) in [(1, 2, 3, 3, 2), (2, 2, 4, 6, 2), (3, 1, 2, 12, 4), (4, 2, 1, 4, 1)]:
    A, B, H = co.gen_batch(700 + seed, nb, p, nx, mb)
    res = []
    for flags in (0, 64):
        h = HipConvexifier(p, nx, mb, chunk=nb, step3=True, flags=flags)
        res.append(h.convexify_step3_batch(A, B, H, 1e-2))
        h.close()
    a, g = res
    print(f'n={nx + mb} p={p}: step3 tuned-vs-generic Hc {rel(g["Hc"], a["Hc"]):.2e} T {rel(g["T"], a["T"]):.2e} iters {a["iters"]} {g["iters"]} status {a["status"]} {g["status"]}', flush=True)
# with rows
p, nx, mb, ng, nc = 3, 5, 2, 2, 3
A, B, H = co.gen_batch(710, 1, p, nx, mb)
rng = np.random.default_rng(5)
J = rng.standard_normal((1, p, ng + nc, nx + mb)); ncnt = np.asarray([[3, 0, 2]], np.int32)
for k in range(p):
    J[0, k, ng + ncnt[0, k]:] = 0.0
res = []
for flags in (0, 64):
    h = HipConvexifier(p, nx, mb, chunk=1, ng=ng, nc=nc, step3=True, flags=flags)
    res.append(h.convexify_step3_con_batch(A, B, H, J, ncnt, 1e-2))
    h.close()
a, g = res
print(f'with rows: Hc {rel(g["Hc"], a["Hc"]):.2e} T {rel(g["T"], a["T"]):.2e} iters {a["iters"]} {g["iters"]} status {a["status"]} {g["status"]}', flush=True)
for (seed, p, nx, mb) in [(300, 2, 24, 10), (301, 2, 30, 10)]:
    A, B, H = co.gen_batch(seed, 1, p, nx, mb)
    t0 = time.time()
    h = HipConvexifier(p, nx, mb, chunk=1, step3=True)
    o = h.convexify_step3_batch(A, B, H, 1e-2)
    h.close()
    t1 = time.time()
    r = co.sdp_step1(A[0], B[0], H[0], rho=1e-2, force=True)
    dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], T=r['T'])[0]
    print(f'n={nx + mb} p={p}: HIP status {o["status"]} iters {o["iters"]} ({t1 - t0:.1f}s) oracle {r["ipm_status"]} iters {r["iters"]} ({time.time() - t1:.1f}s) Hc err {rel(o["Hc"][0], H[0] + dHc):.2e} '
          f'T err {rel(o["T"][0], r["T"]):.2e} kappa {o["kappa"][0]:.8f} vs {r["kappa"]:.8f}', flush=True)
