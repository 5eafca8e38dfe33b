"""One default and one tight step of the bench batch (device-resident), for timing / rocprofv3 --kernel-trace --stats.
    python scripts/tight_timing.py [batch, default 512] [log2(1/tol), default 37]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier  # noqa: E402
from tunempc_amd import synthetic  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lt = int(sys.argv[2]) if len(sys.argv) > 2 else 37
p, nx, mb = 64, 24, 8
nd = min(nb, 512)
A, B, H = synthetic.gen_batch(100000, nd, p, nx, mb)
dev = torch.device('cuda', 0)
dA, dB, dH = (torch.from_numpy(np.ascontiguousarray(x[:nb])).to(dev) for x in (A, B, H))
h = HipConvexifier(p, nx, mb, chunk=0)
out = h.convexify_batch_device(dA, dB, dH, None); torch.cuda.synchronize()
t0 = time.perf_counter(); out = h.convexify_batch_device(dA, dB, dH, out); torch.cuda.synchronize(); t1 = time.perf_counter()
it0 = out['iters'].cpu().numpy().copy(); k0 = out['kappa'].cpu().numpy().copy()
h.set_tight(True, 2.0 ** -lt)
t2 = time.perf_counter(); out = h.convexify_batch_device(dA, dB, dH, out); torch.cuda.synchronize(); t3 = time.perf_counter()
it1 = out['iters'].cpu().numpy(); st = out['status'].cpu().numpy(); k1 = out['kappa'].cpu().numpy()
print(f'batch {nb}: default {t1 - t0:.3f} s ({nb * p / (t1 - t0):.0f} stage-conv/s, {it0.mean():.1f} iterations); tight 2^-{lt}: {t3 - t2:.3f} s ({nb * p / (t3 - t2):.0f} stage-conv/s, '
      f'{it1.mean():.1f} iterations, {int((st == 0).sum())}/{nb} Optimal), extra per added iteration {(t3 - t2 - (t1 - t0)) / (it1.mean() - it0.mean()) * 1e3:.0f} ms; '
      f'kappa drop mean {np.mean(k0 - k1):.3e}')
