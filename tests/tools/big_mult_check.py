"""Multipliers of G / C (Steps 1 and 2) at 32 < n <= 64 -- the <true> forms of k_phi_pre / k_phi_rhs / k_phi_dir on the generic per-stage kernels:
  (a) at n <= 32 against the tuned kernels (debug flag 64), (b) at n > 32 against the numpy oracle.   python tests/tools/big_mult_check.py [cpu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)


def model(seed, nb, p, nx, mb, ng, nc):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 1)
    n = nx + mb
    G = rng.standard_normal((nb, p, ng, n)); C = np.zeros((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            C[b, k, :ncnt[b, k]] = rng.standard_normal((ncnt[b, k], n))
    return A, B, H, G, C, ncnt


def oracle_step2(A, B, H, G, C, ncnt, rho):
    p = A.shape[0]
    Cl = [C[k, :ncnt[k]] if ncnt[k] else None for k in range(p)]
    con = C.shape[1] > 0
    r = co.sdp_step1(A, B, H, G=G if G.shape[1] else None, C=Cl if con else None, rho=rho if con else None)
    dH = co.convex_hessian_suppl(A, B, r['P'], G=G if G.shape[1] else None, Fg=r.get('Fg'), C=Cl if con else None, F=r.get('F'))[0]
    return r, H + dH


def main():
    cpu_only = len(sys.argv) > 1 and sys.argv[1] == 'cpu'
    if not cpu_only:
        from tunempc_amd._lib import HipConvexifier
        for (seed, nb, p, nx, mb, ng, nc) in [(11, 2, 5, 5, 2, 2, 3), (12, 2, 8, 12, 4, 3, 4), (13, 1, 4, 24, 8, 4, 5), (14, 2, 1, 6, 2, 1, 2), (15, 2, 6, 10, 3, 0, 4)]:
            A, B, H, G, C, ncnt = model(seed, nb, p, nx, mb, ng, nc)
            J = np.concatenate([G, C], axis=2)
            res = []
            for flags in (0, 64):
                h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb, flags=flags)
                o2 = h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
                o1 = h.convexify_eq_batch(A, B, H, G) if ng else None
                h.close()
                res.append((o1, o2))
            (a1, a2), (b1, b2) = res
            print(f'n={nx + mb} p={p} ng={ng} nc={nc}: step2 tuned-vs-generic Hc {max(rel(b2["Hc"][b], a2["Hc"][b]) for b in range(nb)):.2e} iters {a2["iters"]} {b2["iters"]} status {a2["status"]} {b2["status"]}'
                  + (f' | eq Hc {max(rel(b1["Hc"][b], a1["Hc"][b]) for b in range(nb)):.2e} iters {a1["iters"]} {b1["iters"]}' if ng else ''), flush=True)
    for (seed, nb, p, nx, mb, ng, nc) in [(31, 1, 3, 8, 4, 20, 18), (32, 1, 2, 20, 10, 24, 24), (33, 1, 3, 12, 6, 31, 0), (21, 1, 3, 24, 10, 2, 3), (22, 1, 4, 20, 16, 3, 2), (23, 1, 2, 30, 12, 0, 4), (24, 1, 3, 26, 8, 2, 0),
                                          (34, 1, 2, 36, 12, 24, 24)]:
        A, B, H, G, C, ncnt = model(seed, nb, p, nx, mb, ng, nc)
        t0 = time.time()
        r, Hc = oracle_step2(A[0], B[0], H[0], G[0], C[0], ncnt[0], 1e-2)
        print(f'n={nx + mb} p={p} ng={ng} nc={nc}: oracle {r["ipm_status"]} iters {r["iters"]} in {time.time() - t0:.1f}s', flush=True)
        if cpu_only:
            continue
        h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb)
        J = np.concatenate([G, C], axis=2)
        if nc:
            o = h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
        else:
            o = h.convexify_eq_batch(A, B, H, G)
        h.close()
        print(f'   HIP status {o["status"]} iters {o["iters"]} Hc err {rel(o["Hc"][0], Hc):.2e} kappa {o["kappa"][0]:.6f} vs {r["kappa"]:.6f}', flush=True)


if __name__ == '__main__':
    main()
