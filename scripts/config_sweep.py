"""Times the BASELINE.json configurations (synthetic data of their shapes) on one GPU; checks invariants."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
from tunempc_amd import synthetic
cfgs = [("c2 unicycle-shaped", 1, 30, 4, 1), ("c3 evaporation-shaped", 256, 50, 2, 2), ("c5 AWE-shaped real size", 64, 40, 9, 6),
        ("c5 synthetic (per-GPU share of 512)", 64, 200, 20, 10), ("c4 bench workload (64 of 512)", 64, 64, 24, 8)]
if len(sys.argv) > 1:
    cfgs = [c for c in cfgs if any(a in c[0] for a in sys.argv[1:])]
out = []
for name, nb, p, nx, mb in cfgs:
    nd = min(nb, 16)
    A, B, H = synthetic.gen_batch(500000, nd, p, nx, mb)
    reps = (nb + nd - 1) // nd
    A, B, H = (np.tile(x, (reps, 1, 1, 1))[:nb] for x in (A, B, H))
    h = HipConvexifier(p, nx, mb, flags=FLAG_PROFILE)
    h.convexify_batch(A, B, H)          # warm-up
    h.profile()
    # short solves (tens of milliseconds) see the clock ramp of an idle part: repeat and keep the median (the minimum is printed as well)
    reps = 1 if nb * p > 20000 else 7
    dts = []
    for _ in range(reps):
        t = time.perf_counter(); o = h.convexify_batch(A, B, H); dts.append(time.perf_counter() - t)
    dt = float(np.median(dts))
    pr = h.profile()
    for k_ in ('factor_ms', 'factor_launches'):
        pr[k_] = pr[k_] / reps
    ev = np.linalg.eigvalsh(o['Hc'])
    ok = bool((ev.min(-1) > 0).all() and ((ev[..., -1] / ev[..., 0]).max(-1) <= o['kappa'] * (1 + 1e-9)).all())
    rec = dict(config=name, batch=nb, p=p, nx=nx, m=mb, seconds_host_to_host=dt, seconds_min=float(min(dts)), repetitions=reps, stage_conv_per_s=nb * p / dt,
               iters_max=int(o['iters'].max()), iters_mean=float(o['iters'].mean()), status=np.bincount(o['status'], minlength=3).tolist(),
               invariants_ok=ok, factor_ms_per_iter=pr['factor_ms'] / max(pr['factor_launches'], 1))
    print(json.dumps(rec)); out.append(rec)
    h.close()
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'config_sweep.json'), 'w'), indent=1)
