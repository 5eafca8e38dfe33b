"""Cycle split of the persistent small-problem kernel (tmpc_persist.h), thread 0 of workgroup 0; needs the -DTMPC_CYCLE_PROF build (built here)."""
import os, sys, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
PROF = os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
if not os.path.exists(PROF) or os.environ.get('REBUILD_PROF'):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value', '-DTMPC_CYCLE_PROF',
                           os.path.join(ROOT, 'tunempc_amd', 'csrc', 'tmpc_api.hip'), '-o', PROF])
L.library_path = lambda: PROF
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
names = ['stage_pre', 'ctrl_a', 'schur assembly', 'factorisation', 'stage_rhs', 'gather', 'substitution', 'border', 'stage_dir', 'eigmin', 'ctrl_b / ctrl_c', 'update + ctrl_d']
shapes = [(1, 30, 4, 1), (256, 50, 2, 2)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]      # python scripts/persist_prof.py nb,p,nx,mb ...
for (nb, p, nx, mb) in shapes:
    A, B, H = synthetic.gen_batch(2000, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=nb)
    h.set_tuning(persistent=2)
    lib = h.lib
    lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
    out = np.zeros(64)
    h.convexify_batch(A, B, H)
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))   # reset
    o = h.convexify_batch(A, B, H)
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
    v = out[48:60]; tot = v.sum(); its = int(o['iters'][0])
    print(f"nb {nb} p {p} nx {nx} mb {mb}: {its} iterations of problem 0, {tot:.3e} cycles in the loop = {tot / its:.0f} per iteration")
    for n_, x in zip(names, v):
        print(f"    {n_:18s} {x:.3e}  {100 * x / max(tot, 1):5.1f} %   {x / its:8.0f} cycles per iteration")
    h.close()
