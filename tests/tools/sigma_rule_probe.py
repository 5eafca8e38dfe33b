"""Experiment (test infrastructure, CPU only): iteration counts of the C++ port at the bench stage shape under other centering-parameter rules of the
Mehrotra corrector (exponent of mu_aff / mu; SDPT3's adaptive exponent) and other caps of the step fraction -- would a different rule save factorisations?
usage: python tests/tools/sigma_rule_probe.py [nb] [p]"""
import os, sys, time, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import cpu_ipm
    from tunempc_amd import synthetic
    nb, p = int(sys.argv[2]), int(sys.argv[3])
    A, B, H = synthetic.gen_batch(100000, nb, p, 24, 8)
    t0 = time.time()
    r = cpu_ipm.convexify_batch(A, B, H, threads=8)
    print(json.dumps(dict(iters=r['iters'].tolist(), status=r['status'].tolist(), kappa=r['kappa'].tolist(), seconds=time.time() - t0)))
    sys.exit(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = int(sys.argv[2]) if len(sys.argv) > 2 else 16
base = None
RULES = [('exponent 2 (product)', {})] + [(f'exponent {e}', {'CPU_IPM_SIGMA_EXP': str(e)}) for e in (1.0, 1.25, 1.5, 1.75, 2.5, 3.0)] + [
    ('SDPT3 adaptive', {'CPU_IPM_SIGMA_EXP': '-1'}), ('exponent 2, gamma <= 0.995', {'CPU_IPM_GAMMA_MAX': '0.995'}), ('exponent 1.5, gamma <= 0.995', {'CPU_IPM_SIGMA_EXP': '1.5', 'CPU_IPM_GAMMA_MAX': '0.995'}),
    ('exponent 2, gamma <= 0.98', {'CPU_IPM_GAMMA_MAX': '0.98'})]
for name, env in RULES:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, __file__, 'child', str(nb), str(p)], env=e, capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(name, 'FAILED', out.stderr[-300:]); continue
    if base is None: base = r
    dk = max(abs(a - b) / b for a, b in zip(r['kappa'], base['kappa']))
    print(f"{name:32s} iterations mean {np.mean(r['iters']):.2f} (min {min(r['iters'])}, max {max(r['iters'])}), Optimal {sum(1 for s_ in r['status'] if s_ == 0)}/{nb}, kappa vs product rule {dk:.1e}, {r['seconds']:.1f} s")
