"""GPU check + batched timing of the register-tile GEMM core (tmpc_gemm2.h) against the first-generation core."""
import sys, json
import numpy as np
sys.path.insert(0, '.')
from tunempc_amd._lib import HipConvexifier

h = HipConvexifier(1, 2, 4, 1)
rng = np.random.default_rng(0)
bad = 0
for (M, N, K, var, mode, tri) in [(304, 304, 304, 1, 0, 0), (304, 304, 304, 1, 1, 0), (304, 304, 304, 1, 2, 0), (304, 304, 304, 1, 0, 1),
                                  (304, 64, 256, 2, 0, 0), (240, 64, 64, 2, 0, 0), (304, 64, 64, 2, 1, 2), (48, 48, 48, 2, 1, 2),
                                  (16, 16, 16, 1, 0, 0), (160, 160, 16, 1, 0, 0), (496, 496, 496, 1, 0, 1), (496, 64, 448, 2, 0, 0),
                                  (320, 160, 32, 1, 2, 0), (336, 96, 64, 2, 0, 0),
                                  (608, 304, 304, 1, 2, 0), (304, 304, 16, 1, 0, 1), (304, 304, 32, 1, 0, 0), (304, 64, 48, 2, 0, 0), (32, 32, 304, 1, 0, 1), (304, 304, 304, 2, 0, 0)]:
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); C0 = rng.standard_normal((M, N))
    if tri == 2:
        B = np.tril(B)
    out = h.debug_gemm_nt(C0, A, B, mode=mode | (var << 4), lower=tri)
    P = A @ B.T
    ref = C0 - P if mode == 0 else (P if mode == 1 else -P)
    if tri == 1:
        nf = M // 16
        mask = np.kron(np.tril(np.ones((nf, nf))), np.ones((16, 16))) > 0
        err = np.abs(out - ref)[mask].max(); keep = np.abs(out - C0)[~mask].max()
    else:
        err = np.abs(out - ref).max(); keep = 0.0
    ok = err < 1e-11 * K and keep == 0.0
    bad += (not ok)
    print(f'gemm2 M={M} N={N} K={K} var={var} mode={mode} tri={tri}: err={err:.2e} untouched={keep:.1e} {"ok" if ok else "FAIL"}')
print('FAILED' if bad else 'all ok')
