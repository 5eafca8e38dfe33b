"""GPU tight-accuracy mode against the C++ CPU restatement (oracle/cpu_ipm, tight=True) on a few small and medium problems.
    python tests/tools/tight_check.py [log2 of 1/tol, default 37]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402
from oracle import cpu_ipm  # noqa: E402
from tunempc_amd import _lib  # noqa: E402

lt = int(sys.argv[1]) if len(sys.argv) > 1 else 37
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for (seed, nb, p, nx, mb) in [(3, 2, 6, 4, 2), (11, 3, 8, 4, 1), (50, 2, 1, 3, 1), (60, 2, 2, 3, 2), (70, 4, 16, 12, 4), (80, 2, 30, 4, 1), (90, 2, 12, 20, 6)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = _lib.HipConvexifier(p, nx, mb, chunk=nb)
    r0 = h.convexify_batch(A, B, H)
    h.set_tight(True, 2.0 ** -lt)
    t0 = time.time(); r = h.convexify_batch(A, B, H); t1 = time.time()
    c = cpu_ipm.convexify_batch(A, B, H, tol=2.0 ** -lt, threads=8, tight=True); t2 = time.time()
    tr = h.trace(nb)
    print(f'seed {seed} nb={nb} p={p} n={nx + mb}: status gpu {r["status"]} cpu {c["status"]} iters gpu {r["iters"]} (default {r0["iters"]}) cpu {c["iters"]} (polish {c["polish_steps"]})')
    print(f'    Hc gpu vs cpu {[float("%.1e" % rel(r["Hc"][i], c["Hc"][i])) for i in range(nb)]}  kappa diff {np.abs(r["kappa"] - c["kappa"]).max():.1e}  mu_t gpu {r["info"][:, 6]} cpu {c["mu_t"]}  '
          f'moved from default {[float("%.1e" % rel(r["Hc"][i], r0["Hc"][i])) for i in range(nb)]}  gpu {t1 - t0:.2f}s cpu {t2 - t1:.2f}s', flush=True)
    h.close()
