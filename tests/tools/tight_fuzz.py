"""Randomised run of the round-4 paths against the C++ CPU restatement (outside pytest):
  (a) tight-accuracy mode, random shapes (n <= 32) and tolerances 2^-29 ... 2^-37;
  (b) stage blocks wider than 32 (generic per-stage kernels), default mode.
Prints every member whose status differs or whose Hc differs by more than 1e-8, and the worst error.
Usage: python tests/tools/tight_fuzz.py [ncases] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402
from oracle import cpu_ipm  # noqa: E402
from tunempc_amd._lib import HipConvexifier  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4040)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
th = max(1, min(16, len(os.sched_getaffinity(0))))
bad = []; worst = {'tight': 0.0, 'big': 0.0}; count = {'tight': 0, 'big': 0}; fellback = 0
t0 = time.time()
for case in range(ncases):
    big = (case % 4 == 3)
    if big:
        nx = int(rng.integers(20, 41)); mb = int(rng.integers(max(1, 33 - nx), min(24, 64 - nx) + 1)); p = int(rng.integers(1, 7))
    else:
        p = int(rng.integers(1, 21)); nx = int(rng.integers(1, 13)) if case % 4 else int(rng.integers(12, 25)); mb = int(rng.integers(1, 9))
    nb = 3
    seed = int(rng.integers(0, 10 ** 6))
    A, B, H = co.gen_batch(seed, nb, p, nx, mb, sigP=float(10.0 ** rng.uniform(-0.5, 1.0)))
    lt = int(rng.integers(29, 38))
    h = HipConvexifier(p, nx, mb, chunk=nb)
    if not big:
        h.set_tight(True, 2.0 ** -lt)
    out = h.convexify_batch(A, B, H)
    h.close()
    ref = cpu_ipm.convexify_batch(A, B, H, tol=(0.0 if big else 2.0 ** -lt), threads=th, tight=not big)
    kind = 'big' if big else 'tight'
    for b in range(nb):
        count[kind] += 1
        rec = dict(case=case, b=b, kind=kind, p=p, nx=nx, mb=mb, seed=seed, lt=lt, status_gpu=int(out['status'][b]), status_cpu=int(ref['status'][b]),
                   iters=int(out['iters'][b]))
        if out['info'][b, 13] != 0.0:
            continue
        if not big and out['info'][b, 6] != ref['mu_t'][b]:
            # the GPU handed the default result back (k_tight_fallback) or the two sides ended at different targets: not a parity statement
            fellback += 1; rec['mu_gpu'] = float(out['info'][b, 6]); rec['mu_cpu'] = float(ref['mu_t'][b]); print('DIFFERENT TARGET', rec); continue
        e = rel(out['Hc'][b], ref['Hc'][b])
        if int(out['status'][b]) != int(ref['status'][b]) or (int(ref['status'][b]) == 0 and e > 1e-8):
            rec['err'] = float(e); bad.append(rec); print('MISMATCH', rec)
        elif int(ref['status'][b]) == 0:
            worst[kind] = max(worst[kind], float(e))
print('members', count, 'worst rel error', worst, 'mismatches', len(bad), 'different target / fell back', fellback, 'seconds %.0f' % (time.time() - t0))
json.dump(dict(count=count, worst=worst, bad=bad, fellback=fellback), open(os.path.join(ROOT, 'gpurun_out', 'tight_fuzz.json'), 'w'), indent=1)
