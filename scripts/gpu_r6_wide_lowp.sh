#!/bin/bash
# round 6 experiment (profiles/r6_wide_block_lowp_experiment.txt; run with a patch that is NOT in the product: O32 carved for n <= 64 at any block width, a conversion kernel behind
# the register-staged k_cr_trsm, the nt <= TRR_NT gate of run_chunk dropped): large-block parity tests, then the large-block shapes with the updates on / off.
# On the product the switch is ignored for these shapes and the two settings time the same.
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tight.py -m gpu -q -p no:cacheprovider -x -k "large or generic or widest or n48 or wide" < /dev/null 2>&1 | tail -n 6
timeout 900 python - <<'PY' 2>&1 | tail -n 12
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
from tunempc_amd import synthetic
for (p, nx, mb, nb) in [(16, 40, 8, 32), (8, 30, 6, 32)]:
    A, B, H = synthetic.gen_batch(300000, nb, p, nx, mb)
    res = {}
    for sw in (0.0, 1e-5, 0.0, 1e-5):
        h = HipConvexifier(p, nx, mb, chunk=nb, flags=FLAG_PROFILE)
        h.set_tuning(lowp_switch=sw)
        h.convexify_batch(A, B, H); h.profile()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); o = h.convexify_batch(A, B, H); ts.append(time.perf_counter() - t0)
        pr = h.profile(); h.close()
        res.setdefault(sw, []).append(min(ts))
        print(f'p={p} n={nx+mb} d={nx*(nx+1)//2} batch {nb} lowp_switch {sw}: {1e3*min(ts):.1f} ms per solve = {nb*p/min(ts):.1f} stage-conv/s, iterations mean {o["iters"].mean():.2f}, Optimal {(o["status"]==0).sum()}/{nb}, lowp factorisations {pr["lowp_factorisations"]:.0f} of {pr["problem_factorisations"]:.0f}')
        res[(sw, 'Hc')] = o['Hc']; res[(sw, 'it')] = o['iters']
    e = np.linalg.norm(res[(1e-5, 'Hc')] - res[(0.0, 'Hc')]) / np.linalg.norm(res[(0.0, 'Hc')])
    print(f'   on vs off: Hc differs by {e:.2e}, iteration counts differ on {(res[(1e-5,"it")] != res[(0.0,"it")]).sum()} of {nb} members; speed-up {min(res[0.0])/min(res[1e-5]):.3f}')
PY
