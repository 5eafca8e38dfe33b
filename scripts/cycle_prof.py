"""In-kernel cycle split of the tile GEMM (wg_gemm_nt) inside the three kernel classes of the block factorisation: thread 0 of block 0
of every launch accumulates __builtin_readcyclecounter() deltas per section; needs the -DTMPC_CYCLE_PROF build (libtunempc_hip_prof.so,
built here when missing)."""
import os, sys, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
PROF = os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
if not os.path.exists(PROF):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value', '-DTMPC_CYCLE_PROF',
                           os.path.join(ROOT, 'tunempc_amd', 'csrc', 'tmpc_api.hip'), '-o', PROF])
L.library_path = lambda: PROF
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(2, 3, 1)
lib = h.lib
lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
out = np.zeros(64)
names = ['issue next-slab loads', 'C prefetch issue', 'ds_read + MFMA', 'wait loads + LDS store', 'epilogue stores', 'barrier']
for (nb, p) in ((512, 64), (64, 64)):
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))   # reset
    ms = h.debug_factor_bench(nb, p, 300, reps=1)
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"nb {nb} p {p} d 300: factor {ms[0]:.2f} ms, solve {ms[1]:.2f} ms (profiled build)")
    v = out[48:51]; tot = v.sum()
    print(f"  k_cr_potrf block column (wave 0 of block 0): {tot:.3e} cycles")
    for n_, x in zip(['left-looking GEMM update', '64 x 64 tile Cholesky + inverse', 'panel multiply by the tile inverse'], v):
        print(f"      {n_:42s} {x:.3e}  {100 * x / max(tot, 1):5.1f} %")
    v = out[56:61]; tot = v.sum()
    print(f"  inside the tile Cholesky: {tot:.3e} cycles")
    for n_, x in zip(['load tile into LDS', 'wave_potrf16 (16 x 16 Cholesky + inverse, one wave)', 'panel + trailing update (MFMA 16x16x4)', 'inverse assembly', 'store tile + inverse'], v):
        print(f"      {n_:52s} {x:.3e}  {100 * x / max(tot, 1):5.1f} %")
    for cls, cn in ((1, 'k_cr_potrf'), (2, 'k_cr_trsm'), (3, 'k_cr_update')):
        v = out[cls * 8: cls * 8 + 6]; tot = v.sum()
        if cls == 2 and os.environ.get('TMPC_TRSM_RR', '1') != '0':
            print(f"  k_cr_trsm_dma: wave 0 of block 0, cycles in the step loop over all launches {tot:.3e}")
            for n_, x in zip(['wait for the slab DMA', 'barrier', 'issue the next DMA', 'ds_read + MFMA', 'tile transitions (store X, park, load E)'], v):
                print(f"      {n_:42s} {x:.3e}  {100 * x / max(tot, 1):5.1f} %")
            continue
        print(f"  {cn}: wave 0 of block 0, cycles inside wg_gemm_nt over all launches {tot:.3e}")
        for n_, x in zip(names, v):
            print(f"      {n_:28s} {x:.3e}  {100 * x / max(tot, 1):5.1f} %")
