"""Where does the fp64 wall at mu ~ 1e-8 come from, and which piece needs more than fp64 to pass it?  (CPU only.)

The oracle's IPM with ONE piece replaced by extended precision (mpmath): the Kronecker-factor images, the assembly of the block-cyclic-
tridiagonal Schur matrix and its factorisation / solves.  Everything else (iterates, residuals, Cholesky / inverses of the n x n cone
blocks, right-hand sides, border, step lengths) stays fp64.  Small problems only (dense (p d)^2 mpmath Cholesky).
    python tests/tools/tight_probe.py"""
import os
import sys
import time

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402

mp.mp.dps = 40


def to_mp(a):
    return mp.matrix(a.tolist())


def T_mp(L, R, ia, ib):
    """co._T in mpmath: T[(ab),(cd)] = <E_ab, L E_cd R'> for the svec basis (weights as Dn' (L kron R) Dn)."""
    d = len(ia)
    out = mp.zeros(d, d)
    for r in range(d):
        a, b = int(ia[r]), int(ib[r])
        for c in range(d):
            cc, dd = int(ia[c]), int(ib[c])
            # Dn' (L kron R) Dn: sum over the (1 or 2) positions of E_ab and E_cd
            rows = [(a, b)] if a == b else [(a, b), (b, a)]
            cols = [(cc, dd)] if cc == dd else [(cc, dd), (dd, cc)]
            s = mp.mpf(0)
            for (i, j) in rows:
                for (k, l) in cols:
                    s += L[i, k] * R[j, l]
            out[r, c] = s
    return out


class MpSystem:
    """dense mpmath Cholesky of the block-cyclic-tridiagonal matrix; solve() takes / returns fp64 arrays like co._CyclicBlockChol"""
    shift = 0.0

    def __init__(self, D, C):
        p = len(D); d = D[0].rows
        self.p, self.d = p, d
        N = p * d
        T = mp.zeros(N, N)
        for k in range(p):
            kn = (k + 1) % p
            for i in range(d):
                for j in range(d):
                    T[k * d + i, k * d + j] += D[k][i, j]
                    if kn == k:
                        T[k * d + i, k * d + j] += C[k][i, j] + C[k][j, i]
                    else:
                        T[k * d + i, kn * d + j] += C[k][i, j]
                        T[kn * d + j, k * d + i] += C[k][i, j]
        self.T = T
        self.L = mp.cholesky(T)

    def solve(self, R):
        p, d = self.p, self.d
        N = p * d
        r = R.reshape(N, -1)
        out = np.zeros_like(r)
        L = self.L
        for q in range(r.shape[1]):
            z = [mp.mpf(float(v)) for v in r[:, q]]
            for i in range(N):
                s = z[i]
                for j in range(i):
                    s -= L[i, j] * z[j]
                z[i] = s / L[i, i]
            for i in range(N - 1, -1, -1):
                s = z[i]
                for j in range(i + 1, N):
                    s -= L[j, i] * z[j]
                z[i] = s / L[i, i]
            out[:, q] = [float(v) for v in z]
        return out.reshape(R.shape)


def assemble_mp(X1, S1i, X2, S2i, V, nx, ia, ib):
    p = X1.shape[0]
    d = len(ia)
    D = [mp.zeros(d, d) for _ in range(p)]
    C = [mp.zeros(d, d) for _ in range(p)]
    for (X, Si) in ((X1, S1i), (X2, S2i)):
        for k in range(p):
            Xm, Sm, Vm = to_mp(X[k]), to_mp(Si[k]), to_mp(V[k])
            Kx = Vm * Xm * Vm.T; Ks = Vm * Sm * Vm.T
            Fx = Xm[:nx, :] * Vm.T; Fs = Sm[:nx, :] * Vm.T
            D[k] += T_mp(Xm[:nx, :nx], Sm[:nx, :nx], ia, ib)
            D[(k + 1) % p] += T_mp(Kx, Ks, ia, ib)
            C[k] -= T_mp(Fx, Fs, ia, ib)
    return D, C


def assemble_round(X1, S1i, X2, S2i, V, nx, ia, ib):
    """extended-precision assembly, rounded to fp64 blocks (factorisation in fp64): does the assembly alone matter?"""
    D, C = assemble_mp(X1, S1i, X2, S2i, V, nx, ia, ib)
    f = lambda M: np.array([[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)])
    return np.stack([f(m) for m in D]), np.stack([f(m) for m in C])


rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)

if __name__ == '__main__':
    cases = [(11, 8, 4, 1), (3, 6, 4, 2)]
    for seed, p, nx, mb in cases:
        A, B, H, _, _ = co.gen_problem(seed, p, nx, mb)
        print(f'seed {seed} p={p} nx={nx} mb={mb}  (N = {2 * p * (nx + mb) + 1})')
        prev = {}
        for lt in (25, 29, 33, 37, 41):
            for name, extra in (('fp64', {}), ('mp', dict(_assemble=assemble_mp, _chol_cls=MpSystem))):
                t0 = time.time()
                r = co.sdp_step1(A, B, H, dict(tol=2.0 ** -lt, max_iter=80, center_iter=20, **extra))
                Hc = H + co.symmetrize(co.calH(A, B, r['P']))
                ev = np.linalg.eigvalsh(Hc)
                msg = f'  tol=2^-{lt} {name:5s}: {r["ipm_status"]:20s} it={r["iters"]:3d} mu_t={r["mu_target"]:.2e} (asked {2.0 ** -lt * r["kappa"]:.2e}) kappa={r["kappa"]:.12f} shift={r["shift"]:.0e} pinf={r["pinf"]:.1e}'
                if name in prev:
                    msg += f'  |Hc - Hc(prev tol)|/|Hc| = {rel(Hc, prev[name][0]):.2e}  dkappa = {r["kappa"] - prev[name][1]:.2e}'
                prev[name] = (Hc, r['kappa'])
                print(msg + f'  [{time.time() - t0:.0f}s]', flush=True)
