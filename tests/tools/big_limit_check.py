"""Blocks beyond 1552 (the compact LDS image of the substitution kernels): plain model at nx = 60 (blocks of 1830) against the C++ port, Step 3 at n = 48 with
24 + 24 rows (blocks of 1893) against the numpy oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
import convexify_oracle as co
import cpu_ipm
from big_mult_check import model
from tunempc_amd._lib import HipConvexifier

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)

for (seed, p, nx, mb) in [(401, 2, 60, 4), (402, 3, 63, 1)]:
    A, B, H = co.gen_batch(seed, 1, p, nx, mb)
    t0 = time.time()
    h = HipConvexifier(p, nx, mb, chunk=1)
    o = h.convexify_batch(A, B, H)
    h.close()
    t1 = time.time()
    c = cpu_ipm.convexify_batch(A, B, H, threads=16)
    print(f'plain nx={nx} n={nx + mb} p={p}: HIP status {o["status"]} iters {o["iters"]} ({t1 - t0:.1f}s) cpu {c["status"]} {c["iters"]} ({time.time() - t1:.1f}s) Hc err {rel(o["Hc"][0], c["Hc"][0]):.2e} '
          f'kappa {o["kappa"][0]:.8f} vs {c["kappa"][0]:.8f}', flush=True)
p, nx, mb, ng, nc = 2, 36, 12, 24, 24
A, B, H, G, C, ncnt = model(34, 1, p, nx, mb, ng, nc)
t0 = time.time()
h = HipConvexifier(p, nx, mb, chunk=1, ng=ng, nc=nc, step3=True)
o = h.convexify_step3_con_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 1e-2)
h.close()
t1 = time.time()
print(f'step3 n=48 rows 24+24: HIP status {o["status"]} iters {o["iters"]} ({t1 - t0:.1f}s)', flush=True)
Cl = [C[0, k, :ncnt[0, k]] if ncnt[0, k] else None for k in range(p)]
r = co.sdp_step1(A[0], B[0], H[0], G=G[0], C=Cl, rho=1e-2, force=True)
dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], G=G[0], Fg=r['Fg'], C=Cl, F=r['F'], T=r['T'])[0]
print(f'   oracle {r["ipm_status"]} iters {r["iters"]} ({time.time() - t1:.1f}s) Hc err {rel(o["Hc"][0], H[0] + dHc):.2e} T err {rel(o["T"][0], r["T"]):.2e} kappa {o["kappa"][0]:.8f} vs {r["kappa"]:.8f}', flush=True)
