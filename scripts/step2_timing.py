"""Cost of the stage-local multipliers on the benchmark shape (n = 32, p = 64): plain Step 1, Step 1 with G (ng = 2) and the
Step 2 model (ng = 2, up to 4 rows of C_k, ragged) on the same 512 problems.  Usage: python scripts/step2_timing.py [nb]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p, nx, mb, ng, nc = 64, 24, 8, 2, 4
n = nx + mb
A, B, H = synthetic.gen_batch(100000, min(nb, 64), p, nx, mb)
rep = (nb + A.shape[0] - 1) // A.shape[0]
A, B, H = (np.concatenate([x] * rep)[:nb] for x in (A, B, H))
rng = np.random.default_rng(1)
G = rng.standard_normal((nb, p, ng, n))
C = rng.standard_normal((nb, p, nc, n))
ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
for b in range(nb):
    for k in range(p):
        C[b, k, ncnt[b, k]:] = 0.0
h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc, flags=1 << 1 if False else 0)
res = {}
def run(name, f):
    f()                                   # warm-up
    t = time.perf_counter(); out = f(); dt = time.perf_counter() - t
    res[name] = dict(seconds=dt, stage_conv_per_s=nb * p / dt, iters_max=int(out['iters'].max()), iters_mean=float(out['iters'].mean()),
                     optimal=int((out['status'] == 0).sum()), kappa_mean=float(out['kappa'].mean()))
    print(name, json.dumps(res[name]))
run('step1', lambda: h.convexify_batch(A, B, H))
run('step1_with_G', lambda: h.convexify_eq_batch(A, B, H, G))
run('step2_G_and_C', lambda: h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 1e-3))
res['config'] = dict(nb=nb, p=p, nx=nx, mb=mb, ng=ng, nc_max=nc, note='host-buffer entries (PCIe copies included), one chunk')
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/step2_timing.json', 'w'), indent=1)
