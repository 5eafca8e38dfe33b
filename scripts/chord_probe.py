"""Chord (frozen-factorisation) centering steps: iterations, factorisations, time and agreement with the all-Newton answer for several thresholds."""
import os, sys, time, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
    from tunempc_amd import synthetic
    nb = int(sys.argv[2])
    A, B, H = synthetic.gen_batch(100000, nb, 64, 24, 8)
    h = HipConvexifier(64, 24, 8, chunk=nb, flags=FLAG_PROFILE)
    h.set_tuning(chord_step=float(sys.argv[4]))      # (rounds 2-3: the environment variable TMPC_CHORD, removed from the library in round 4)
    h.convexify_batch(A[:8], B[:8], H[:8]); h.profile()
    t = time.time(); o = h.convexify_batch(A, B, H); t = time.time() - t
    pr = h.profile(); tr = h.trace(min(nb, 4))
    np.save(sys.argv[3], o['Hc'])
    cent = [[(round(r[1], 2), '%.1e' % r[8], '%.1e' % r[7]) for r in tr[b] if r[0] > 0 and r[1] >= 1] for b in range(min(nb, 2))]
    print(json.dumps(dict(t=t, iters_max=int(o['iters'].max()), iters_mean=float(o['iters'].mean()), optimal=int((o['status'] == 0).sum()),
                          problem_factorisations=pr['problem_factorisations'], phases=pr['factor_launches'], centering=cent)))
    sys.exit(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ref = None
for thr in ('0', '1000', '100', '30', '10', '4'):
    f = f'/tmp/chord_{thr}.npy'
    out = subprocess.run([sys.executable, __file__, 'child', str(nb), f, thr], capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(thr, 'FAILED', out.stdout[-500:], out.stderr[-800:]); continue
    Hc = np.load(f)
    if ref is None: ref = Hc
    err = max(np.linalg.norm(Hc[b] - ref[b]) / np.linalg.norm(ref[b]) for b in range(nb))
    print(f"chord_step {thr:>5}: {r['t']*1e3:8.1f} ms  iters max {r['iters_max']} mean {r['iters_mean']:.2f}  optimal {r['optimal']}/{nb}  problem-factorisations {r['problem_factorisations']:.0f} "
          f"({r['problem_factorisations']/nb:.2f} per problem)  max rel diff vs all-Newton {err:.2e}")
    print('      centering steps of problems 0,1 (phase[.25 = chord], step norm, raw step):', r['centering'])
