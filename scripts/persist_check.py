"""Round 5: the persistent one-launch interior-point loop (tmpc_persist.h) against the launch-sequence path on small shapes: values, iteration counts, time per solve.
   python scripts/persist_check.py [quick]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic

shapes = [(1, 30, 4, 1), (4, 1, 3, 1), (3, 2, 2, 2), (5, 3, 5, 3), (8, 7, 4, 2), (16, 12, 5, 1), (256, 50, 2, 2), (64, 30, 4, 1), (2, 160, 3, 2), (4, 33, 1, 1), (512, 30, 4, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    shapes = shapes[:4]
if len(sys.argv) > 1 and sys.argv[1] == 'short':              # the p <= 16 half of the rule at larger batches
    shapes = [(nb, p, nx, mb) for (p, nx, mb) in ((8, 4, 1), (16, 3, 2), (4, 5, 3)) for nb in (1, 8, 32, 64, 94)]
if len(sys.argv) > 1 and sys.argv[1] == 'sweep':             # where the persistent kernel starts to pay (the handle splits a batch over two lanes: half of nb each)
    shapes = [(nb, p, nx, mb) for (p, nx, mb) in ((30, 4, 1), (50, 2, 2), (20, 3, 2), (100, 3, 1)) for nb in (64, 96, 128, 192, 256, 384)]
for (nb, p, nx, mb) in shapes:
    A, B, H = synthetic.gen_batch(2000, nb, p, nx, mb)
    res = {}
    for mode in (0, 1):
        h = HipConvexifier(p, nx, mb, chunk=nb)
        h.set_tuning(persistent=2 * mode)      # 0: launch sequence, 2: persistent kernel whatever the batch
        h.convexify_batch(A, B, H)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); out = h.convexify_batch(A, B, H); ts.append(time.perf_counter() - t)
        res[mode] = (out, sorted(ts)[2])
        h.close()
    o0, t0 = res[0]; o1, t1 = res[1]
    rel = np.linalg.norm((o0['Hc'] - o1['Hc']).reshape(nb, -1), axis=1) / np.linalg.norm(o0['Hc'].reshape(nb, -1), axis=1)
    print(f"nb {nb:4d} p {p:3d} nx {nx} mb {mb}: sequence {t0*1e3:8.3f} ms  persistent {t1*1e3:8.3f} ms  ({t0/t1:5.2f}x)  {nb*p/t1/1e3:9.1f} k stage-conv/s   "
          f"max rel dHc {rel.max():.2e}  kappa rel {np.max(np.abs(o0['kappa']-o1['kappa'])/np.abs(o0['kappa'])):.1e}  iters {o0['iters'].max()}/{o1['iters'].max()} "
          f"equal {bool((o0['iters']==o1['iters']).all())}  status {np.bincount(o1['status'], minlength=3).tolist()} same {bool((o0['status']==o1['status']).all())}", flush=True)
