"""Robustness sweep of the HIP path outside the benchmark distribution: scale of the hidden P (how indefinite H is),
conditioning of the hidden SPD target, spectral radius of A, shapes.  Checks status and the solver-independent invariants
(Hc > 0, cond(Hc_k) <= kappa, Hc - H = sym(calH(P)))."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
if os.environ.get('TMPC_LIB'):
    L.library_path = lambda: os.path.join(ROOT, 'tunempc_amd', 'lib', os.environ['TMPC_LIB'])
from tunempc_amd._lib import HipConvexifier


from tunempc_amd import synthetic


def gen(seed, p, nx, mb, sigP, cond_exp, rad):
    return synthetic.gen_problem(seed, p, nx, mb, sigP=sigP, cond_exp=cond_exp, rad=rad)


def calH(A, B, P):
    V = np.concatenate([A, B], axis=2); nx = A.shape[1]
    d = np.swapaxes(V, 1, 2) @ np.roll(P, -1, axis=0) @ V
    d[:, :nx, :nx] -= P
    return (d + np.swapaxes(d, 1, 2)) / 2


CONDS = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 3, 5)      # exponents of cond(Hhat)
rows = []
nbad = 0
SHAPES = [(1, 3, 1), (2, 4, 2), (5, 9, 6), (30, 4, 1), (8, 16, 4)]
if os.environ.get('ROBUST_WIDE'):      # round 6: only shapes whose Schur blocks are wide enough for the single-precision updates (TMPC_TUNE_LOWP_SWITCH: blocks wider than 64)
    SHAPES = [(6, 24, 8), (12, 12, 4)]
for (p, nx, mb) in SHAPES:
    h = HipConvexifier(p, nx, mb)
    if os.environ.get('ROBUST_LOWP'):         # switch-over of the single-precision updates (0 = fp64 throughout)
        h.set_tuning(lowp_switch=float(os.environ['ROBUST_LOWP']))
    if os.environ.get('ROBUST_PERSIST'):      # round 5: 2 = the persistent one-launch kernel for every small shape whatever the batch, 0 = never (tmpc_set_tuning)
        h.set_tuning(persistent=int(os.environ['ROBUST_PERSIST']))
    for sigP in (0.1, 1.0, 10.0, 100.0):
        for cond_exp in CONDS:
            for rad in (0.5, 0.9, 1.2):
                nb = 8
                ABH = [gen(7000 + 17 * b, p, nx, mb, sigP, cond_exp, rad) for b in range(nb)]
                A = np.stack([x[0] for x in ABH]); B = np.stack([x[1] for x in ABH]); H = np.stack([x[2] for x in ABH])
                out = h.convexify_batch(A, B, H)
                ok = 0
                for b in range(nb):
                    early = bool(out['info'][b, 13])
                    ev = np.linalg.eigvalsh(out['Hc'][b])
                    struct = np.abs(out['Hc'][b] - H[b] - calH(A[b], B[b], out['P'][b])).max() / max(1.0, np.abs(H[b]).max())
                    cond = (ev[:, -1] / ev[:, 0]).max()
                    good = out['status'][b] == 0 and ev.min() > 0 and struct < 1e-10 and (early or cond <= out['kappa'][b] * (1 + 1e-7))
                    ok += bool(good)
                rows.append(dict(p=p, nx=nx, mb=mb, sigP=sigP, cond_exp=cond_exp, rad=rad, ok=ok, nb=nb, iters_max=int(out['iters'].max()),
                                 status=np.bincount(out['status'], minlength=3).tolist(), kappa_max=float(out['kappa'].max())))
                if ok != nb:
                    nbad += 1
                    print('NOT ALL OK', rows[-1])
    h.close()
print('cases', len(rows), 'with a failing member', nbad, 'max iterations', max(r['iters_max'] for r in rows))
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', os.environ.get('ROBUST_OUT', 'robustness_sweep.json')), 'w'))
