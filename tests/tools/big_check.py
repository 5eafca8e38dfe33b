"""The generic per-stage kernels (tmpc_big.h): (a) forced at n <= 32 (debug flag 64) against the tuned kernels and the CPU port, (b) at n > 32 against the CPU port.
    python tests/tools/big_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from oracle import cpu_ipm
from tunempc_amd import _lib
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for (seed, nb, p, nx, mb) in [(3, 2, 6, 4, 2), (70, 3, 16, 12, 4), (95, 2, 5, 24, 8), (60, 2, 2, 3, 2), (61, 2, 1, 5, 2)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h0 = _lib.HipConvexifier(p, nx, mb, chunk=nb); r0 = h0.convexify_batch(A, B, H); h0.close()
    h1 = _lib.HipConvexifier(p, nx, mb, chunk=nb, flags=64); t0 = time.time(); r1 = h1.convexify_batch(A, B, H); t1 = time.time(); h1.close()
    print(f'n={nx + mb} p={p}: generic vs tuned Hc {[float("%.1e" % rel(r1["Hc"][i], r0["Hc"][i])) for i in range(nb)]} status {r1["status"]} {r0["status"]} iters {r1["iters"]} {r0["iters"]} kappa diff {np.abs(r1["kappa"] - r0["kappa"]).max():.1e} ({t1 - t0:.2f}s)', flush=True)
for (seed, nb, p, nx, mb) in [(201, 2, 4, 30, 10), (202, 2, 6, 36, 12), (203, 1, 3, 40, 24), (204, 2, 8, 33, 1)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = _lib.HipConvexifier(p, nx, mb, chunk=nb); t0 = time.time(); r = h.convexify_batch(A, B, H); t1 = time.time(); h.close()
    c = cpu_ipm.convexify_batch(A, B, H, threads=8); t2 = time.time()
    print(f'n={nx + mb} (nx={nx}) p={p}: GPU vs cpu_ipm Hc {[float("%.1e" % rel(r["Hc"][i], c["Hc"][i])) for i in range(nb)]} status {r["status"]} {c["status"]} iters {r["iters"]} {c["iters"]} kappa diff {np.abs(r["kappa"] - c["kappa"]).max():.1e} (gpu {t1 - t0:.2f}s cpu {t2 - t1:.2f}s)', flush=True)
