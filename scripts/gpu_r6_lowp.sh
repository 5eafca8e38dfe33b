#!/bin/bash
# round 6: literal fragment ranges in the update tiles + single-precision Schur-complement updates in the early main-phase iterations (TMPC_TUNE_LOWP_SWITCH):
# bit-identity of the fp64 path, unit check of the float32 tile, isolated factorisation timing, whole solves
export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/r6_lowp_check.txt
python - > $OUT 2>&1 <<'PY'
import sys, time, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
from tunempc_amd import synthetic
import cpu_ipm
def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
# (0) the fp64 path computes what it computed before (digests of round 6's first GPU run, library of commit 5761969)
want = {(6, 64, 24, 8): 'c78a7be20cb1c9af', (3, 20, 20, 10): '87e548e8d7bd7ee6', (4, 9, 9, 6): '25469faf0d8bdb8c', (5, 7, 12, 4): '2319f3966e133705'}
for (nb, p, nx, mb), dg in want.items():
    A, B, H = synthetic.gen_batch(31337, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=nb); h.set_tuning(lowp_switch=0.0)
    o = h.convexify_batch(A, B, H); h.close()
    got = hashlib.sha256(o['Hc'].tobytes() + o['P'].tobytes() + o['iters'].tobytes()).hexdigest()[:16]
    print('fp64 path digest', (nb, p, nx, mb), got, 'OK' if got == dg else 'CHANGED (was %s)' % dg, flush=True)
# (1) block solve with every update in single precision against numpy (unit check of wg_tile_dma_f32 + the float32 copies of k_cr_trsm_dma)
rng = np.random.default_rng(5)
for (p, d) in ((3, 40), (5, 100), (4, 300), (7, 304), (9, 210), (3, 72)):
    D = np.zeros((p, d, d)); Cc = rng.standard_normal((p, d, d)) * 0.05
    for k in range(p):
        m = rng.standard_normal((d, d)); D[k] = m @ m.T / d + 2.0 * np.eye(d)
    T = np.zeros((p * d, p * d))
    for k in range(p):
        T[k*d:(k+1)*d, k*d:(k+1)*d] += D[k]; kn = (k + 1) % p
        T[k*d:(k+1)*d, kn*d:(kn+1)*d] += Cc[k]; T[kn*d:(kn+1)*d, k*d:(k+1)*d] += Cc[k].T
    rhs = rng.standard_normal((p, d))
    ref = np.linalg.solve(T, rhs.ravel()).reshape(p, d)
    errs = []
    for sw in (0.0, 2.0):
        h = HipConvexifier(2, 3, 1); h.set_tuning(lowp_switch=sw)
        errs.append(rel(h.debug_block_solve(D, Cc, rhs)[0], ref)); h.close()
    print(f'block solve p={p} d={d}: rel. error fp64 updates {errs[0]:.2e}, fp32 updates {errs[1]:.2e}', flush=True)
# (2) isolated factorisation timing
for sw in (0.0, 2.0, 0.0, 2.0):
    h = HipConvexifier(2, 3, 1); h.set_tuning(lowp_switch=sw)
    for (nb, p, d) in ((433, 64, 300), (512, 64, 300)):
        ms = h.debug_factor_bench(nb, p, d, reps=3)
        fl = nb * (max(p - 2, 0) * 6.3333 * d ** 3 + 2.3333 * d ** 3 + d ** 3 / 3)
        print(f'factor bench, updates {"fp32" if sw else "fp64"}: nb {nb} p {p} d {d}: factor {ms[0]:8.3f} ms ({fl / ms[0] / 1e9:6.2f} TF/s algorithmic)  solve {ms[1]:7.3f} ms', flush=True)
    h.close()
# (3) whole solves at the bench shape: default vs switch values; parity of every member against cpu_ipm (fp64), iterations, timing
nb, p, nx, mb = 128, 64, 24, 8
A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
ref = cpu_ipm.convexify_batch(A, B, H, threads=16)
base = None
for sw in (0.0, 1e-4, 3e-5, 1e-5, 3e-6, 0.0, 1e-5):
    h = HipConvexifier(p, nx, mb, chunk=nb); h.set_tuning(lowp_switch=sw)
    h.convexify_batch(A[:4], B[:4], H[:4])
    h.set_options(flags=FLAG_PROFILE); h.profile()
    t0 = time.perf_counter(); o = h.convexify_batch(A, B, H); t = time.perf_counter() - t0
    pf = h.profile(); h.close()
    e = max(rel(o['Hc'][b], ref['Hc'][b]) for b in range(nb))
    if base is None: base = o
    print(f'switch {sw:7.0e}: {nb * p / t:8.0f} stage-conv/s  iterations {o["iters"].mean():.2f} (max {o["iters"].max()})  optimal {(o["status"] == 0).sum()}/{nb}  worst vs cpu_ipm {e:.2e}  vs switch 0 {max(rel(o["Hc"][b], base["Hc"][b]) for b in range(nb)):.2e}'
          f'  fp32-update factorisations/problem {pf["lowp_factorisations"] / nb:.2f} of {pf["problem_factorisations"] / nb:.2f}  update {pf["update_ms"]:.0f} ms trsm {pf["trsm_ms"]:.0f} potrf {pf["potrf_ms"]:.0f} factor {pf["factor_ms"]:.0f} pass1 {pf["pass1_ms"]:.0f} total {pf["total_ms"]:.0f}', flush=True)
PY
cat $OUT | cut -c 1-420
