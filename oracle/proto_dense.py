"""Scratch prototype: dense (unstructured) primal-dual IPM for the convexifier SDP.
TEST INFRASTRUCTURE ONLY. Used to validate the structured oracle on small cases.
"""
import numpy as np


def gen_problem(seed, p, nx, nu, sigP=1.0, identity=False):
    rng = np.random.default_rng(seed)
    n = nx + nu
    A = np.zeros((p, nx, nx)); B = np.zeros((p, nx, nu)); H = np.zeros((p, n, n))
    Phat = np.zeros((p, nx, nx)); Hhat = np.zeros((p, n, n))
    for k in range(p):
        a = rng.standard_normal((nx, nx)) / np.sqrt(nx)
        rho = np.max(np.abs(np.linalg.eigvals(a)))
        A[k] = a * (0.9 / rho)
        B[k] = rng.standard_normal((nx, nu)) / np.sqrt(nx)
        W, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = 10.0 ** rng.uniform(0, 1, n)
        if identity:
            lam[:] = 1.0
        Hhat[k] = (W * lam) @ W.T
        pk = rng.standard_normal((nx, nx)); Phat[k] = sigP * (pk + pk.T) / 2
    for k in range(p):
        V = np.hstack([A[k], B[k]])
        dH = V.T @ Phat[(k + 1) % p] @ V
        dH[:nx, :nx] -= Phat[k]
        H[k] = Hhat[k] - dH
        H[k] = (H[k] + H[k].T) / 2
    return A, B, H, Phat, Hhat


def svec_basis(nx):
    """list of symmetric basis matrices E_ab (a<=b), unnormalised: e_a e_b^T + e_b e_a^T (a!=b), e_a e_a^T"""
    bas = []
    for a in range(nx):
        for b in range(a, nx):
            E = np.zeros((nx, nx))
            E[a, b] = 1.0; E[b, a] = 1.0
            bas.append(E)
    return bas


def build_maps(A, B, Hbar):
    """Return list of (Lmat (n*n, m), C (n,n)) for the 2p LMI blocks: S_j(y) = C_j + L_j y.
    y = (tau, alpha, P_0.., P_{p-1})."""
    p, nx, _ = A.shape
    n = Hbar.shape[1]
    d = nx * (nx + 1) // 2
    m = 2 + p * d
    bas = svec_basis(nx)
    blocks = []
    I = np.eye(n)
    for k in range(p):
        V = np.hstack([A[k], B[k]])
        LM = np.zeros((n * n, m))
        LM[:, 1] = Hbar[k].ravel()
        kn = (k + 1) % p
        for i, E in enumerate(bas):
            LM[:, 2 + kn * d + i] += (V.T @ E @ V).ravel()
            Ee = np.zeros((n, n)); Ee[:nx, :nx] = E
            LM[:, 2 + k * d + i] -= Ee.ravel()
        L1 = LM.copy()
        L2 = -LM.copy(); L2[:, 0] = I.ravel()
        blocks.append((L1, -I))
        blocks.append((L2, np.zeros((n, n))))
    return blocks, m, d


def sym(M):
    return 0.5 * (M + M.T)


def max_step(X, dX, frac=1.0):
    """largest theta in (0, inf) with X + theta dX >= 0"""
    Lc = np.linalg.cholesky(X)
    Li = np.linalg.inv(Lc)
    w = np.linalg.eigvalsh(Li @ dX @ Li.T)
    lm = w.min()
    if lm >= 0:
        return np.inf
    return -1.0 / lm


def solve(A, B, H, tol=1e-8, maxit=60, verbose=True, alpha_min=1e-8, mu_final=None, center_its=6):
    p, nx, _ = A.shape
    n = H.shape[1]
    ev = np.concatenate([np.linalg.eigvalsh(H[k]) for k in range(p)])
    aev = np.abs(ev); aev = aev[aev != 0]
    s = 1.0 / aev.min(); sbeta = aev.max() / aev.min()
    Hbar = s * H
    blocks, m, d = build_maps(A, B, Hbar)
    J = len(blocks)
    N = J * n + 1
    c = np.zeros(m); c[0] = 1.0
    # initial point
    y = np.zeros(m); y[0] = 2 * sbeta; y[1] = 1.0
    I = np.eye(n)

    def Sof(y):
        return [(L @ y).reshape(n, n) + C for (L, C) in blocks]

    Sy = Sof(y)
    S = []
    X = []
    for j in range(J):
        w = np.linalg.eigvalsh(Sy[j])
        if w.min() > 0.1 * sbeta:
            S.append(Sy[j].copy())
        else:
            S.append(sbeta * I.copy())
        X.append(I / (p * n))
    s0 = max(y[1] - alpha_min, 1.0)
    x0 = 1.0 / (p * n)
    hist = []
    phase = 0
    mu_t = None
    ncent = 0
    for it in range(maxit):
        Sy = Sof(y)
        Rd = [Sy[j] - S[j] for j in range(J)]
        rd0 = (y[1] - alpha_min) - s0
        mu = (sum(np.sum(X[j] * S[j]) for j in range(J)) + x0 * s0) / N
        # dual residual r_p = c - sum L^T X
        LX = sum(blocks[j][0].T @ X[j].ravel() for j in range(J))
        LX[1] += x0
        rp = c - LX
        pinf = np.linalg.norm(rp) / (1 + np.linalg.norm(c))
        dinf = np.sqrt(sum(np.sum(r * r) for r in Rd) + rd0 ** 2) / (1 + np.sqrt(sum(np.sum(q * q) for q in S)))
        gap = N * mu
        relgap = gap / max(1.0, abs(y[0]))
        if verbose:
            print(f"it {it:2d} tau={y[0]:.10f} alpha={y[1]:.4e} mu={mu:.3e} pinf={pinf:.2e} dinf={dinf:.2e} relgap={relgap:.2e} ph={phase}")
        hist.append((y[0], mu, pinf, dinf))
        if phase == 0 and relgap < tol and pinf < tol and dinf < tol:
            phase = 1
            if mu_final is None:
                mu_t = tol * max(1.0, abs(y[0])) / N
                mu_t = 2.0 ** np.round(np.log2(mu_t))
            else:
                mu_t = mu_final
        if phase == 1:
            if ncent >= center_its:
                break
            ncent += 1
        Sinv = [np.linalg.inv(S[j]) for j in range(J)]
        # Schur
        Bm = np.zeros((m, m))
        for j in range(J):
            L = blocks[j][0]
            K = np.kron(X[j], Sinv[j])  # vec(X dS Sinv) = (Sinv^T kron X) vec... use symmetric form
            K = 0.5 * (np.kron(X[j], Sinv[j]) + np.kron(Sinv[j], X[j]))
            Bm += L.T @ K @ L
        Bm[1, 1] += x0 / s0
        Bm = sym(Bm)
        reg=0.0
        while True:
            try:
                cf = np.linalg.cholesky(Bm + reg*np.diag(np.diag(Bm)))
                break
            except np.linalg.LinAlgError:
                reg = 1e-14 if reg==0 else reg*100
                if verbose: print('   chol fail -> reg',reg)

        def lin_solve(r):
            z = np.linalg.solve(cf, r)
            return np.linalg.solve(cf.T, z)

        def direction(sig_mu, corr=None, corr0=0.0):
            # rhs = sum L^T (sig_mu Sinv - X Rd Sinv [- corr]) - c   (plus scalar block)
            rhs = -c.copy()
            for j in range(J):
                T = sig_mu * Sinv[j] - sym(X[j] @ Rd[j] @ Sinv[j])
                if corr is not None:
                    T = T - corr[j]
                rhs += blocks[j][0].T @ T.ravel()
            t0 = sig_mu / s0 - x0 * rd0 / s0 - corr0
            rhs[1] += t0
            dy = lin_solve(rhs)
            dS = [(blocks[j][0] @ dy).reshape(n, n) + Rd[j] for j in range(J)]
            dX = []
            for j in range(J):
                T = sig_mu * Sinv[j] - X[j] - sym(X[j] @ dS[j] @ Sinv[j])
                if corr is not None:
                    T = T - corr[j]
                dX.append(T)
            ds0 = dy[1] + rd0
            dx0 = sig_mu / s0 - x0 - x0 * ds0 / s0 - corr0
            return dy, dS, dX, ds0, dx0

        def steps(dS, dX, ds0, dx0):
            ap = min(max_step(X[j], dX[j]) for j in range(J))
            ad = min(max_step(S[j], dS[j]) for j in range(J))
            if dx0 < 0: ap = min(ap, -x0 / dx0)
            if ds0 < 0: ad = min(ad, -s0 / ds0)
            return ap, ad

        if phase == 0:
            dy, dS, dX, ds0, dx0 = direction(0.0)
            ap, ad = steps(dS, dX, ds0, dx0)
            ap = min(1.0, ap); ad = min(1.0, ad)
            mu_aff = (sum(np.sum((X[j] + ap * dX[j]) * (S[j] + ad * dS[j])) for j in range(J)) + (x0 + ap * dx0) * (s0 + ad * ds0)) / N
            sigma = (mu_aff / mu) ** 3
            sigma = min(max(sigma, 1e-8), 1.0)
            corr = [sym(dX[j] @ dS[j] @ Sinv[j]) for j in range(J)]
            corr0 = dx0 * ds0 / s0
            dy, dS, dX, ds0, dx0 = direction(sigma * mu, corr, corr0)
            ap, ad = steps(dS, dX, ds0, dx0)
            gam = 0.9 + 0.09 * min(ap, ad) if min(ap, ad) < 1e30 else 0.99
            ap = min(1.0, gam * ap); ad = min(1.0, gam * ad)
        else:
            dy, dS, dX, ds0, dx0 = direction(mu_t)
            ap, ad = steps(dS, dX, ds0, dx0)
            ap = min(1.0, 0.95 * ap); ad = min(1.0, 0.95 * ad)
            if verbose:
                print(f"   center: |dy|={np.linalg.norm(dy):.3e} ap={ap:.3f} ad={ad:.3f}")
        for j in range(J):
            X[j] = sym(X[j] + ap * dX[j])
            S[j] = sym(S[j] + ad * dS[j])
        x0 += ap * dx0; s0 += ad * ds0
        y = y + ad * dy
    alpha = y[1]; tau = y[0]
    Pbar = np.zeros((p, nx, nx))
    bas = svec_basis(nx)
    for k in range(p):
        for i, E in enumerate(bas):
            Pbar[k] += y[2 + k * d + i] * E
    Pst = Pbar / (s * alpha)
    Hc = np.zeros_like(H)
    for k in range(p):
        V = np.hstack([A[k], B[k]])
        dH = V.T @ Pst[(k + 1) % p] @ V
        dH[:nx, :nx] -= Pst[k]
        Hc[k] = H[k] + sym(dH)
    return dict(Hc=Hc, P=Pst, alpha=alpha, tau=tau, beta=tau / sbeta, it=it, s=s, sbeta=sbeta, y=y)


if __name__ == "__main__":
    import sys
    p, nx, nu = 3, 3, 2
    A, B, H, Phat, Hhat = gen_problem(0, p, nx, nu)
    r = solve(A, B, H)
    conds = [np.linalg.cond(r['Hc'][k]) for k in range(p)]
    print("conds", conds, "tau", r['tau'], "cond Hhat", [np.linalg.cond(Hhat[k]) for k in range(p)])
    print("min eig", [np.linalg.eigvalsh(r['Hc'][k]).min() for k in range(p)])
