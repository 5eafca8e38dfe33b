"""First-contact GPU diagnostics: building blocks, then full solves vs the oracle.  Writes gpurun_out/gpu_check.log"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier, FLAG_NO_MFMA, FLAG_PROFILE

np.set_printoptions(linewidth=200, precision=6)
rng = np.random.default_rng(0)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check_gemm(h):
    for (M, N, K) in [(64, 64, 16), (64, 64, 64), (128, 64, 48), (304, 48, 256), (112, 112, 112), (304, 304, 304)]:
        A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); Cm = rng.standard_normal((M, N))
        for mode, ref in [(0, Cm - A @ B.T), (1, A @ B.T), (2, -A @ B.T)]:
            out = h.debug_gemm_nt(Cm, A, B, mode)
            print(f"gemm M{M} N{N} K{K} mode{mode}: rel err {rel(out, ref):.2e}")
        if M == N:
            out = h.debug_gemm_nt(Cm, A, A, 0, lower=True)
            ref = Cm - A @ A.T
            il = np.tril_indices(M)
            # lower mode works on 64x64 tiles: compare the tile-lower part
            mask = (np.arange(M)[:, None] // 64) >= (np.arange(N)[None, :] // 64)
            print(f"  syrk-lower: rel err {rel(out[mask], ref[mask]):.2e}; untouched upper ok {np.array_equal(out[~mask], Cm[~mask])}")


def cyclic_dense(D, Cc):
    p, d, _ = D.shape
    T = np.zeros((p * d, p * d))
    for k in range(p):
        T[k*d:(k+1)*d, k*d:(k+1)*d] += D[k]
        kn = (k + 1) % p
        if kn == k:
            T[k*d:(k+1)*d, k*d:(k+1)*d] += Cc[k] + Cc[k].T
        else:
            T[k*d:(k+1)*d, kn*d:(kn+1)*d] += Cc[k]
            T[kn*d:(kn+1)*d, k*d:(k+1)*d] += Cc[k].T
    return T


def check_block_solve(h):
    for (p, d) in [(1, 6), (2, 10), (3, 10), (5, 21), (4, 78), (3, 136), (3, 300), (6, 45)]:
        # random SPD block-cyclic-tridiagonal matrix  T = sum_k J_k' J_k + I,  J_k = [E_k F_k] on blocks (k, k+1)
        E = rng.standard_normal((p, 2 * d, d)) / np.sqrt(2 * d); Fm = rng.standard_normal((p, 2 * d, d)) / np.sqrt(2 * d)
        D = np.stack([np.eye(d) for _ in range(p)]); Cc = np.zeros((p, d, d))
        for k in range(p):
            D[k] += E[k].T @ E[k]; D[(k + 1) % p] += Fm[k].T @ Fm[k]; Cc[k] = E[k].T @ Fm[k]
        T = cyclic_dense(D, Cc)
        ev = np.linalg.eigvalsh(T)
        rhs = rng.standard_normal((p, d))
        xref = np.linalg.solve(T, rhs.ravel()).reshape(p, d)
        x, ns = h.debug_block_solve(D, Cc, rhs)
        print(f"block solve p{p} d{d}: min eig {ev[0]:.3f} rel err {rel(x, xref):.2e} nshift {ns}")


def check_eig(h, p, n):
    H = rng.standard_normal((3, p, n, n)); H = H + H.transpose(0, 1, 3, 2)
    out = h.eig_scan(H)
    ev = np.linalg.eigvalsh(H)
    ref = np.stack([ev[..., 0], ev[..., -1], np.abs(ev).min(-1), np.abs(ev).max(-1)], axis=-1)
    print(f"eig scan n{n}: max abs err {np.abs(out - ref).max():.2e}")


def check_solve(p, nx, mb, nb, seed, flags=0, verbose=True):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, flags=flags | FLAG_PROFILE)
    t = time.time()
    out = h.convexify_batch(A, B, H)
    dt = time.time() - t
    prof = h.profile()
    errs = []; 
    for b in range(min(nb, 3)):
        r = co.convexify_arrays(A[b], B[b], H[b])
        e = rel(out['Hc'][b], r['Hc'])
        errs.append(e)
        if verbose:
            print(f"  b{b}: gpu kappa {out['kappa'][b]:.10f} it {out['iters'][b]} st {out['status'][b]} | oracle kappa {r['kappa']:.10f} it {r['iters']} st {r['status']} | rel Hc err {e:.2e}")
    print(f"solve p{p} nx{nx} mb{mb} nb{nb} flags{flags}: {dt:.3f}s  max rel err {max(errs):.2e}  status counts {np.bincount(out['status'], minlength=3)} iters max {out['iters'].max()}")
    print("   info[0]:", out['info'][0])
    print("   profile:", {k: round(v, 2) for k, v in prof.items()})
    h.close()
    return out


if __name__ == '__main__':
    which = sys.argv[1:] or ['blocks', 'small', 'mid']
    h0 = HipConvexifier(4, 4, 2)
    print(h0.lib.tmpc_version().decode())
    if 'blocks' in which:
        h0.set_options(flags=FLAG_NO_MFMA); print('--- scalar-FMA fragments'); check_gemm(h0)
        h0.set_options(flags=0); print('--- MFMA'); check_gemm(h0)
        check_block_solve(h0)
        for n in (4, 5, 15, 32):
            hh = HipConvexifier(3, n - 1, 1); check_eig(hh, 3, n); hh.close()
    if 'small' in which:
        check_solve(3, 3, 2, 2, 0)
        check_solve(1, 3, 1, 2, 20)
        check_solve(2, 3, 1, 2, 30)
        check_solve(30, 4, 1, 2, 13)
        check_solve(16, 3, 2, 4, 5)
    if 'mid' in which:
        check_solve(6, 12, 4, 4, 11)
        check_solve(4, 24, 8, 2, 12)
        check_solve(4, 24, 8, 2, 12, flags=FLAG_NO_MFMA)
    if 'big' in which:
        check_solve(64, 24, 8, 8, 100, verbose=False)
