"""Experiment: two half-batches on two streams/handles from two host threads, with a start offset."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
p, nx, mb = 64, 24, 8
nbl = 512
nd = 64
A, B, H = synthetic.gen_batch(100000, nd, p, nx, mb)
dev = torch.device('cuda', 0)
tile = lambda x: torch.from_numpy(np.tile(x, (nbl // nd, 1, 1, 1)).copy()).to(dev)
dA, dB, dH = tile(A), tile(B), tile(H)
h1 = HipConvexifier(p, nx, mb, chunk=256); h2 = HipConvexifier(p, nx, mb, chunk=256)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
halves = [(h1, s1, slice(0, 256)), (h2, s2, slice(256, 512))]
outs = [None, None]

def run(i, delay):
    h, s, sl = halves[i]
    time.sleep(delay)
    outs[i] = h.convexify_batch_device(dA[sl].contiguous(), dB[sl].contiguous(), dH[sl].contiguous(), outs[i], stream=s.cuda_stream)

for delay in (0.0, 0.1, 0.2, 0.3):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(i, delay * i)) for i in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"two streams, offset {delay:.1f}s: {dt:.3f} s  -> {nbl*p/dt:.0f} stage-conv/s   status ok {(outs[0]['status']==0).sum().item() + (outs[1]['status']==0).sum().item()}")
h = HipConvexifier(p, nx, mb, chunk=512)
o = None
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); o = h.convexify_batch_device(dA, dB, dH, o); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"single stream: {dt:.3f} s -> {nbl*p/dt:.0f} stage-conv/s")
