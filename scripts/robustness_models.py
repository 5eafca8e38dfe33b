"""Hard targets (cond(Hhat) = 1e5) through the models with stage-local multipliers: Step 1 with G, Step 2 (G and active rows of C with
norm terms), Step 3 (T with its norm cone).  The plain model has the wide sweep (robustness_sweep.py); this one checks that the back-off
logic of round 3 behaves there as well: finite outputs, a status, Hc > 0 whenever the status is not Infeasible, and the counts."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tunempc_amd import synthetic
from tunempc_amd._lib import HipConvexifier

TIGHT = len(sys.argv) > 1 and sys.argv[1] == 'tight'      # round 5: the same members in the tight-accuracy mode (Steps 1 and 2; members whose continuation fails fall back: counted)
rows = []
for (p, nx, mb) in [(5, 9, 6), (30, 4, 1), (8, 6, 3)]:
    n = nx + mb
    for cond_exp in (3, 5):
        for rad in (0.5, 0.9):
            nb = 8
            probs = [synthetic.gen_problem(9000 + 13 * b, p, nx, mb, sigP=10.0, cond_exp=cond_exp, rad=rad) for b in range(nb)]
            A, B, H = (np.stack([q[i] for q in probs]) for i in range(3))
            rng = np.random.default_rng(p * 100 + cond_exp)
            ng, nc = 2, 2
            G = rng.standard_normal((nb, p, ng, n)); Cc = rng.standard_normal((nb, p, nc, n))
            ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
            for b in range(nb):
                for k in range(p):
                    Cc[b, k, ncnt[b, k]:] = 0.0
            h = HipConvexifier(p, nx, mb, ng=ng, nc=nc)
            if TIGHT:
                h.set_tight(True)
            outs = dict(G=h.convexify_eq_batch(A, B, H, G), step2=h.convexify_step2_batch(A, B, H, np.concatenate([G, Cc], axis=2), ncnt, 1e-3))
            h.close()
            if n <= 12 and not TIGHT:
                h3 = HipConvexifier(p, nx, mb, step3=True)
                outs['step3'] = h3.convexify_step3_batch(A, B, H, 1e-3)
                h3.close()
            for model, o in outs.items():
                fin = all(np.isfinite(o[k]).all() for k in ('Hc', 'P', 'kappa'))
                pd = [bool(np.linalg.eigvalsh(o['Hc'][b]).min() > 0) for b in range(nb)]
                okpd = all(pd[b] for b in range(nb) if o['status'][b] != 2)
                mut0 = 2.0 ** np.round(np.log2(2.0 ** -25 * np.maximum(1.0, o['kappa'])))
                back = np.where(o['info'][:, 13] != 0, 0, np.round(np.log2(np.maximum(o['info'][:, 6], 1e-300) / mut0))).astype(int)
                rows.append(dict(p=p, nx=nx, mb=mb, cond_exp=cond_exp, rad=rad, model=model, status=np.bincount(o['status'], minlength=3).tolist(),
                                 finite=bool(fin), pd_ok=bool(okpd), iters_max=int(o['iters'].max()), backoffs=back.tolist(), tight=TIGHT,
                                 fell_back_to_default=int((o['info'][:, 10] == 4.0).sum()), mu_target_min=float(o['info'][:, 6].min()), mu_target_max=float(o['info'][:, 6].max())))
                print(rows[-1], flush=True)
print('rows', len(rows), 'not finite', sum(not r['finite'] for r in rows), 'PD violated', sum(not r['pd_ok'] for r in rows),
      'Optimal', sum(r['status'][0] for r in rows), 'Feasible', sum(r['status'][1] for r in rows), 'Infeasible', sum(r['status'][2] for r in rows),
      'fell back to the default point', sum(r['fell_back_to_default'] for r in rows))
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'robustness_models_tight.json' if TIGHT else 'robustness_models.json'), 'w'))
