"""Assembled Schur rows of the Step 3 variables at the first iteration (init state) against a numpy assembly with the oracle's formulas."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
from tunempc_amd._lib import HipConvexifier, cr_schedule
seed, p, nx, mb, rho = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (300, 2, 2, 1, 1e-3)
n = nx + mb; d = nx * (nx + 1) // 2; m = n * (n + 1) // 2
A, B, H = co.gen_batch(seed, 1, p, nx, mb)
A, B, H = A[0], B[0], co.symmetrize(H[0])
s, sbeta = co.auto_scaling(H); Hb = s * H; alpha = 1.0 / sbeta; tau = 2.0; x0 = 1.0 / (p * n)
I = np.eye(n); V = np.concatenate([A, B], axis=2)
X = [np.broadcast_to(x0 * I, (p, n, n)).copy() for _ in range(2)]
S = [np.broadcast_to(I, (p, n, n)).copy(), tau * I - alpha * Hb]
Si = [np.linalg.inv(S[0]), np.linalg.inv(S[1])]
wr = rho * sbeta / s
ta, tb = np.triu_indices(n); cw = np.where(ta == tb, 1.0, np.sqrt(2.0)); we = np.where(ta == tb, 1.0, 2.0)
ph = min(1.0, x0 * n / wr); z = x0 / ph; an = wr * ph * n; t0 = 0.5 * (x0 + np.sqrt(x0 * x0 + 4 * an * an))
sv = np.concatenate([[t0], wr * cw * ph]); xv = x0 * co._soc_inv(sv)
beta, v = co._soc_scaling(sv, xv); W2 = co._soc_W2inv(beta, v)
GG = np.zeros((m, n, n))
for e in range(m):
    GG[e, ta[e], tb[e]] = 1.0; GG[e, tb[e], ta[e]] = 1.0
ia, ib = np.triu_indices(nx)
dp = (d + m + 1 + 15) // 16 * 16
sched = cr_schedule(p)
h = HipConvexifier(p, nx, mb, step3=True, chunk=1, flags=8)
h.convexify_step3_batch(A[None], B[None], H[None], rho)
D = h.debug_array(3, 0, p * dp * dp).reshape(p, dp, dp)
O = h.debug_array(5, 0, p * dp * dp).reshape(p, dp, dp)
W3 = h.debug_array(7, 0, p * dp * 3).reshape(p, dp, 3)
worst = {}
def upd(name, got, ref):
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
    worst[name] = max(worst.get(name, 0.0), err)
    return err
for k in range(p):
    kn = (k + 1) % p
    Wm = sum(co.symmetrize(X[r][k][None] @ GG @ Si[r][k][None]) for r in range(2))           # [m, n, n]
    a = -np.stack([co._svec_grad(Wm[e][:nx, :nx], ia, ib) for e in range(m)])                # [m, d]
    b = np.stack([co._svec_grad(V[k] @ Wm[e] @ V[k].T, ia, ib) for e in range(m)])
    Hth = np.einsum('vab,wab->vw', GG, Wm); Hth = 0.5 * (Hth + Hth.T) + np.diag(np.full(m, z / ph)) + np.outer(wr * cw, wr * cw) * W2[1:, 1:]
    got_b = D[kn][d:d + m, :d] if p > 1 else None
    if p > 1:
        print('stage', k, 'b-coupling err', upd('b', got_b, b))
        got_a = O[k][d:d + m, :d] if sched['orient'][k] == 0 else O[k][:d, d:d + m].T
        print('stage', k, 'a-coupling err', upd('a', got_a, a), 'orient', sched['orient'][k])
    else:
        print('stage', k, 'a+b err', upd('ab', D[kn][d:d + m, :d], a + b))
    got_H = np.tril(D[kn][d:d + m, d:d + m])
    print('stage', k, 'theta-theta err', upd('H', got_H, np.tril(Hth)))
    print('stage', k, 't row err', upd('t', D[kn][d + m, d:d + m + 1], np.concatenate([wr * cw * W2[0, 1:], [W2[0, 0]]])))
    # right-hand side (pass 1: sigma = 0) and border entries
    Y = [None, None]
    Mk = alpha * Hb[k] + ph * np.ones((n, n))
    T1 = -co.symmetrize(X[0][k] @ ((Mk - I) - S[0][k]) @ Si[0][k]); T2 = -co.symmetrize(X[1][k] @ ((tau * I - Mk) - S[1][k]) @ Si[1][k])
    g = -xv
    r_th = we * (T1 - T2)[ta, tb] + 0.0 + wr * cw * (g[1:] + xv[1:]); r_t = g[0] + xv[0] - 1.0
    Psi = co.symmetrize(X[1][k] @ Si[1][k]); Phi = co.symmetrize(X[0][k] @ Hb[k] @ Si[0][k]) + co.symmetrize(X[1][k] @ Hb[k] @ Si[1][k])
    print('stage', k, 'rhs err', upd('rhs', W3[kn][d:d + m + 1, 0], np.concatenate([r_th, [r_t]])),
          'c_tau err', upd('ct', W3[kn][d:d + m, 1], -we * Psi[ta, tb]), 'c_alpha err', upd('ca', W3[kn][d:d + m, 2], we * Phi[ta, tb]))
print(worst)
np.set_printoptions(precision=4, linewidth=200)
print('D[1] rows d..d+3:'); print(D[1][d:d+4, :d+6])
kk=0; Wm = sum(co.symmetrize(X[r][kk][None] @ GG @ Si[r][kk][None]) for r in range(2)); print('expected b rows:'); print(np.stack([co._svec_grad(V[kk] @ Wm[e] @ V[kk].T, ia, ib) for e in range(4)]))
print('Ddiag', h.debug_array(2, 0, p*dp).reshape(p, dp)[1][:d+8])
