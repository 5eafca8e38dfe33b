"""Round 6, VERDICT r5 item 2c -- oracle-first experiment: what would an fp32 block factorisation of the Schur complement in the EARLY main-phase iterations cost / buy?
(MEASUREMENT ONLY; the product and the oracle's default path are untouched.)

The HIP path factors the block-cyclic-tridiagonal HKM Schur matrix in fp64 on the matrix cores (78.6 TFLOP/s peak).  fp32 MFMA has twice that rate and half the
bytes in the factorisation and the substitutions.  Here the numpy oracle's factorisation is replaced, while mu / max(1, tau) > switch, by one whose blocks and
factors are rounded to float32 (LAPACK spotrf / strsm on float32 copies), followed by `refine` steps of iterative refinement against the fp64 blocks
(r = b - T x in fp64, x += solve32(r)).  Reported per switch-over value: main-phase + centering iterations, how many factorisations were fp32, how far the returned
point is from the all-fp64 answer.  The kill criterion of the review: fewer than 4 factorisations per problem moved to fp32, or any member above 1e-9.

usage: python tests/tools/fp32_factor_probe.py [nprob]"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import convexify_oracle as co  # noqa: E402


def make_cls(trace, switch, refine, count):
    class Chol32(co._CyclicBlockChol):
        def __init__(self, D, C):
            mu = trace[-1]['mu']; tau = trace[-1]['tau']
            self.low = mu / max(1.0, abs(tau)) > switch
            if not self.low:
                count['f64'] += 1
                super().__init__(D, C)
                return
            count['f32'] += 1
            self.D64, self.C64 = D, C
            p, d, _ = D.shape
            self.p, self.d = p, d
            self.shift = 0.0
            try:
                self._factor32(D.astype(np.float32), C.astype(np.float32))
            except np.linalg.LinAlgError:
                self.low = False
                count['f32'] -= 1; count['f64'] += 1; count['fallback'] += 1
                super().__init__(D, C)

        def _factor32(self, D, C):
            p, d = self.p, self.d
            if p <= 2:
                raise np.linalg.LinAlgError('p <= 2: not probed')
            Lkk = np.zeros((p, d, d), np.float32); O = np.zeros((p, d, d), np.float32); F = np.zeros((p, d, d), np.float32)
            Dw = D.copy()
            Fpre = C[p - 1].copy()
            tr = lambda L, B: sla.solve_triangular(L, B.T, lower=True).T.astype(np.float32)          # B L^-T in float32
            for k in range(p - 1):
                Lkk[k] = np.linalg.cholesky(Dw[k])
                sub = C[k].T.copy()
                if k == p - 2:
                    sub = sub + Fpre
                    O[k] = tr(Lkk[k], sub)
                    Dw[p - 1] -= O[k] @ O[k].T
                else:
                    O[k] = tr(Lkk[k], sub); F[k] = tr(Lkk[k], Fpre)
                    Dw[k + 1] -= O[k] @ O[k].T
                    Dw[p - 1] -= F[k] @ F[k].T
                    Fpre = -F[k] @ O[k].T
            Lkk[p - 1] = np.linalg.cholesky(Dw[p - 1])
            assert Lkk.dtype == np.float32 and O.dtype == np.float32
            self.Lkk, self.O, self.F = Lkk, O, F

        def _apply(self, X):                       # T X in fp64 from the fp64 blocks
            D, C, p = self.D64, self.C64, self.p
            Y = np.einsum('kab,kbr->kar', D, X)
            for k in range(p):
                kn = (k + 1) % p
                Y[k] += C[k] @ X[kn]
                Y[kn] += C[k].T @ X[k]
            return Y

        def solve(self, R):
            if not self.low:
                return super().solve(R)
            X = super().solve(R.astype(np.float32)).astype(np.float64)
            for _ in range(refine):
                X = X + super().solve((R - self._apply(X)).astype(np.float32)).astype(np.float64)
            return X
    return Chol32


def main():
    nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    shapes = [(8, 4, 1), (12, 6, 3), (16, 12, 4)]
    print('# fp32 block factorisation in the early main phase (oracle experiment).  columns: switch-over mu/kappa, refinement steps; per shape: iterations (all-fp64 baseline),')
    print('#   factorisations in fp32 / fp64 per problem, fp32 Cholesky failures, worst |Hc - Hc_fp64| (rel. Frobenius), members not Optimal')
    base = {}
    for (p, nx, mb) in shapes:
        for b in range(nprob):
            A, B, H = co.gen_problem(52000 + 31 * b + p, p, nx, mb)[:3]
            r = co.convexify_arrays(A, B, H)
            base[(p, nx, mb, b)] = (A, B, H, r)
    for switch in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        for refine in (0, 1, 2):
            line = f'switch {switch:7.0e} refine {refine}: '
            for (p, nx, mb) in shapes:
                its, its0, worst, bad = [], [], 0.0, 0
                count = dict(f32=0, f64=0, fallback=0)
                for b in range(nprob):
                    A, B, H, r0 = base[(p, nx, mb, b)]
                    trace = []
                    r = co.sdp_step1(A, B, H, dict(_chol_cls=make_cls(trace, switch, refine, count)), trace=trace)
                    Hc = H + co.symmetrize(co.calH(A, B, r['P']))
                    worst = max(worst, np.linalg.norm(Hc - r0['Hc']) / np.linalg.norm(r0['Hc']))
                    bad += r['ipm_status'] != 'optimal'
                    its.append(r['iters']); its0.append(r0['iters'])
                line += f' | p={p} n={nx + mb}: it {np.mean(its):5.2f} ({np.mean(its0):5.2f})  f32 {count["f32"] / nprob:4.1f} f64 {count["f64"] / nprob:4.1f} fail {count["fallback"]}  dHc {worst:.1e} bad {bad}'
            print(line, flush=True)


if __name__ == '__main__':
    main()
