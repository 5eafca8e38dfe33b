"""Round 6, VERDICT r5 item 2c at the BENCH shape: oracle/cpu_ipm with the Schur-complement UPDATES of the block factorisation (D_s -= O_s O_s', fill = -O_a O_b': the
work of k_cr_update_dma, 36 % of a step) computed from float32 roundings of the O blocks with float32 accumulation while mu / kappa > switch; Cholesky, triangular
solves and every substitution stay fp64, no refinement.  (CPU_IPM_LOWP_SWITCH: experiment hook of cpu_ipm.cpp, test infrastructure.)
usage: python tests/tools/fp32_update_probe.py nb p nx mb [threads]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import cpu_ipm
    from tunempc_amd import synthetic
    nb, p, nx, mb, th = (int(v) for v in sys.argv[2:7])
    A, B, H = synthetic.gen_batch(100000, nb, p, nx, mb)
    o = cpu_ipm.convexify_batch(A, B, H, threads=th, tight=False) if not os.environ.get('CPU_IPM_LOWP_SWITCH') else None
    lib = cpu_ipm.load()
    import ctypes as C
    Hc = np.empty_like(H); kappa = np.empty(nb); status = np.empty(nb, np.int32); iters = np.empty(nb, np.int32); info = np.zeros((nb, 4))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib.cpu_ipm_convexify_batch2(nb, p, nx, mb, d(A), d(B), d(H), 0.0, th, 0, d(Hc), d(kappa), i(status), i(iters), d(info))
    np.savez(sys.argv[7], Hc=Hc, kappa=kappa, status=status, iters=iters, lowp=info[:, 1])
    sys.exit(0)

nb, p, nx, mb = (int(v) for v in sys.argv[1:5])
th = int(sys.argv[5]) if len(sys.argv) > 5 else 8
res = {}
for sw in ('0', '3e-5', '1e-5'):
    env = dict(os.environ)
    if sw != '0':
        env['CPU_IPM_LOWP_SWITCH'] = sw
    else:
        env.pop('CPU_IPM_LOWP_SWITCH', None)
    out = f'/tmp/fp32u_{sw}.npz'
    subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', str(nb), str(p), str(nx), str(mb), str(th), out], env=env)
    res[sw] = np.load(out)
b0 = res['0']
print(f'# {nb} problems of the bench generator, p={p} nx={nx} m={mb}; fp32 updates while mu/kappa > switch; columns: iterations mean (max), fp32-update factorisations per problem,')
print('#   worst / median rel. Frobenius distance of Hc to the all-fp64 answer, worst |kappa - kappa64|/kappa, members not Optimal')
for sw, r in res.items():
    e = np.array([np.linalg.norm(r['Hc'][b] - b0['Hc'][b]) / np.linalg.norm(b0['Hc'][b]) for b in range(nb)])
    print(f'switch {sw:>5}: iterations {r["iters"].mean():6.2f} ({r["iters"].max()})  fp32 {(r["lowp"] % 1000).mean():5.2f} (pivot failures {int((r["lowp"] // 1000).sum())})  dHc worst {e.max():.1e} median {np.median(e):.1e}  dkappa {np.abs(r["kappa"] / b0["kappa"] - 1).max():.1e}  not optimal {(r["status"] != 0).sum()}', flush=True)
