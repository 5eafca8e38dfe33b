"""How well are the SECONDARY outputs (P, Fg, F, T) determined?  (the GPU tests hold them to 1e-7 / 1e-6 while Hc is held to 1e-8.)

For every case of tests/test_gpu_parity.py that asserts one of them: the oracle is solved twice, on inputs that differ by 1e-14 relative
(rounding-level noise).  The ratio  (relative change of P) / (relative change of Hc)  is a lower bound of the condition number of the map
output-Hc -> P: two computations that agree on Hc to 1e-10 cannot be expected to agree on P better than that ratio times 1e-10.  With
`--gpu` the HIP path is compared with the oracle on the same cases and the worst errors are printed next to it.
    python tests/tools/secondary_tol_probe.py [--gpu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
GPU = '--gpu' in sys.argv
if GPU:
    import torch  # noqa: F401
    from tunempc_amd._lib import HipConvexifier
import convexify_oracle as co  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def noisy(H, seed):
    rng = np.random.default_rng(seed)
    E = rng.standard_normal(H.shape) * 1e-14
    return H * (1.0 + co.symmetrize(E))


rows = []


def report(kind, case, pairs, gpu_pairs=None):
    dh = pairs['Hc']
    line = f'{kind:7s} {str(case):42s} oracle self-reproducibility: Hc {dh:.1e}'
    for k, v in pairs.items():
        if k != 'Hc':
            line += f' | {k} {v:.1e} (x{v / max(dh, 1e-17):.0f})'
    if gpu_pairs:
        line += '  || GPU vs oracle: ' + ', '.join(f'{k} {v:.1e}' for k, v in gpu_pairs.items())
    print(line, flush=True)
    rows.append((kind, case, pairs, gpu_pairs))


# ---- plain model: P (test_golden_vectors asserts 1e-7 on P)
for seed, nb, p, nx, mb in [(0, 3, 3, 3, 2), (13, 2, 30, 4, 1), (5, 4, 16, 3, 2), (11, 2, 6, 12, 4), (12, 2, 4, 24, 8)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    out = HipConvexifier(p, nx, mb).convexify_batch(A, B, H) if GPU else None
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b])
        if r['early_exit']:
            continue
        r2 = co.convexify_arrays(A[b], B[b], noisy(H[b], b))
        g = dict(Hc=rel(out['Hc'][b], r['Hc']), P=rel(out['P'][b], r['P'])) if GPU else None
        report('plain', (seed, b, p, nx, mb), dict(Hc=rel(r2['Hc'], r['Hc']), P=rel(r2['P'], r['P'])), g)

# ---- Step 1 with G: Fg
for seed, nb, p, nx, mb, ng in [(20, 2, 3, 3, 2, 2), (7, 2, 5, 4, 2, 3), (11, 2, 4, 6, 3, 4), (5, 2, 8, 8, 4, 8)]:
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    G = np.random.default_rng(1000 + seed).standard_normal((nb, p, ng, nx + mb))
    out = HipConvexifier(p, nx, mb, ng=ng).convexify_eq_batch(A, B, H, G) if GPU else None
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b], G=G[b])
        if r['early_exit']:
            continue
        r2 = co.convexify_arrays(A[b], B[b], noisy(H[b], b), G=G[b])
        g = dict(Hc=rel(out['Hc'][b], r['Hc']), Fg=rel(out['Fg'][b], r['Fg']), P=rel(out['P'][b], r['P'])) if GPU else None
        report('eq', (seed, b, p, nx, mb, ng), dict(Hc=rel(r2['Hc'], r['Hc']), Fg=rel(r2['Fg'], r['Fg']), P=rel(r2['P'], r['P'])), g)

# ---- Step 3: T
for seed, nb, p, nx, mb, rho in [(0, 2, 2, 2, 1, 1e-3), (1, 2, 3, 3, 2, 1e-2), (6, 1, 4, 4, 3, 1.0), (4, 1, 2, 9, 6, 1e-3), (5, 1, 5, 2, 2, 1e-1)]:
    A, B, H = co.gen_batch(300 + seed, nb, p, nx, mb)
    out = HipConvexifier(p, nx, mb, step3=True).convexify_step3_batch(A, B, H, rho) if GPU else None
    for b in range(nb):
        if min(np.linalg.eigvalsh(co.symmetrize(H[b, k])).min() for k in range(p)) > 0:
            continue

        def solve(Hx):
            r = co.sdp_step1(A[b], B[b], Hx, rho=rho, force=True)
            dHc = co.check_convergence(A[b], B[b], Hx, r['P'], r['ipm_status'], T=r['T'])[1]
            return r, Hx + dHc
        r, Hc = solve(H[b]); r2, Hc2 = solve(noisy(H[b], b))
        g = dict(Hc=rel(out['Hc'][b], Hc), T=rel(out['T'][b], r['T']), P=rel(out['P'][b], r['P'])) if GPU else None
        report('step3', (seed, b, p, nx, mb, rho), dict(Hc=rel(Hc2, Hc), T=rel(r2['T'], r['T']), P=rel(r2['P'], r['P'])), g)

print()
for key in ('P', 'Fg', 'T'):
    amp = [pr[key] / max(pr['Hc'], 1e-17) for _, _, pr, _ in rows if key in pr]
    worst = max(pr[key] for _, _, pr, _ in rows if key in pr)
    line = f'{key}: amplification over Hc (oracle vs oracle, inputs 1e-14 apart) median x{np.median(amp):.0f}, max x{max(amp):.0f}; worst self-reproducibility {worst:.1e}'
    if GPU:
        line += f'; worst GPU vs oracle {max(gp[key] for _, _, _, gp in rows if gp and key in gp):.1e}'
    print(line)
