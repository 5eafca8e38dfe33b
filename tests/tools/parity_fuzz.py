"""Randomised parity run (outside pytest): random shapes and seeds, plain Step 1 / Step 1 with G / Step 2 model / Step 2 with the beta-only
objective (rho = 0) / Step 3 model (n <= 12), HIP path vs the oracle.  Prints every member whose status differs or whose Hc differs by more than 1e-8, and the worst error per model.
Usage: python tests/tools/parity_fuzz.py [ncases] [seed] [pmax] [nxmax]"""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
pmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12
nxmax = int(sys.argv[4]) if len(sys.argv) > 4 else 8
MODELS = ('plain', 'G', 'step2', 'beta', 'step3')
worst = {m: 0.0 for m in MODELS}; bad = []; count = {m: 0 for m in MODELS}
t0 = time.time()
for case in range(ncases):
    p = int(rng.integers(1, pmax + 1)); nx = int(rng.integers(1, nxmax + 1)); mb = int(rng.integers(1, 5))      # mb = 0 (a stage block without inputs) is degenerate: kappa* = 1 with both LMIs active everywhere
    n = nx + mb
    ng = int(rng.integers(1, 4)); nc = int(rng.integers(1, 5))
    seed = int(rng.integers(0, 10 ** 6))
    nb = 3
    A, B, H = co.gen_batch(seed, nb, p, nx, mb, sigP=float(10.0 ** rng.uniform(-0.5, 1.0)))
    gs = float(10.0 ** rng.uniform(-1, 1))
    G = gs * rng.standard_normal((nb, p, ng, n)); C = gs * rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            C[b, k, ncnt[b, k]:] = 0.0
    rho = float(10.0 ** rng.uniform(-3, 0))
    h = HipConvexifier(p, nx, mb, ng=ng, nc=nc)
    outs = dict(plain=h.convexify_batch(A, B, H), G=h.convexify_eq_batch(A, B, H, G),
                step2=h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, rho))
    outs['beta'] = h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 0.0)
    h.close()
    do3 = n <= 12
    if do3:
        h3 = HipConvexifier(p, nx, mb, step3=True)
        outs['step3'] = h3.convexify_step3_batch(A, B, H, rho)
        h3.close()
    for b in range(nb):
        early = np.linalg.eigvalsh(H[b])[:, 0].min() > 0
        Cl = [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        for model in MODELS:
            if model == 'step3' and not do3:
                continue
            o = outs[model]
            if early:
                ok = bool(o['info'][b, 13]) and not o['dHc'][b].any()
                err, st = (0.0 if ok else 1.0), 0
            else:
              try:
                if model == 'plain':
                    r = co.convexify_arrays(A[b], B[b], H[b]); Hc, st = r['Hc'], r['status']
                elif model == 'G':
                    r = co.convexify_arrays(A[b], B[b], H[b], G=G[b]); Hc, st = r['Hc'], r['status']
                elif model == 'step3':
                    r = co.sdp_step1(A[b], B[b], H[b], rho=rho, force=True)
                    st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], T=r['T'])[:2]
                    Hc = H[b] + dHc
                else:
                    r = co.sdp_step1(A[b], B[b], H[b], G=G[b], C=Cl, rho=rho, cost_free=(model == 'beta'))
                    st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], G=G[b], Fg=r['Fg'], C=Cl, F=r['F'])[:2]
                    Hc = H[b] + dHc
              except np.linalg.LinAlgError as e:
                print('ORACLE FAILED', dict(case=case, b=b, model=model, p=p, nx=nx, mb=mb, seed=seed, err=str(e), status_gpu=int(o['status'][b])))
                continue
              err = np.linalg.norm(o['Hc'][b] - Hc) / np.linalg.norm(Hc)
            count[model] += 1
            worst[model] = max(worst[model], err if int(o['status'][b]) == int(st) == 0 else 0.0)
            if int(o['status'][b]) != int(st) or (int(st) == 0 and err > 1e-8):
                bad.append(dict(case=case, b=b, model=model, p=p, nx=nx, mb=mb, ng=ng, nc=nc, seed=seed, gs=gs, rho=rho, err=float(err),
                                status_gpu=int(o['status'][b]), status_oracle=int(st), iters=int(o['iters'][b])))
                print('MISMATCH', bad[-1])
print('members', count, 'worst rel error among Optimal/Optimal', worst, 'mismatches', len(bad), 'seconds %.0f' % (time.time() - t0))
json.dump(dict(count=count, worst=worst, bad=bad), open(os.path.join(ROOT, 'gpurun_out', 'parity_fuzz.json'), 'w'), indent=1)
