"""CPU restatement of the COMPLETE reference SDP -- Steps 1, 2 and 3 of `convexify()` with the G / C / T terms --
as a plain dense model + a small generic interior-point solver.  TEST INFRASTRUCTURE ONLY (imported by tests/ and by
nothing under tunempc_amd/).  Parity unpinned: the reference solves this model with PICOS -> CVXOPT/MOSEK, neither
of which exists in this image; the model below is pinned to the reference by construction (every variable,
constraint and objective term cites its line) and checked in tests/ against the structured Step-1 oracle, against
constructed known-answer cases and through the invariants of SURVEY.md section 8c.

What it restates (tunempc/convexifier.py):
  * `setUpModelPicos`        :213-308   variables alpha, beta, dP_i (symmetric nx x nx), Fg_i (ng), F_i (nc_i), T_i
                                        (symmetric n x n); alpha > 1e-8; Fg, F, T > 0 elementwise; objective
                                        beta + sum ||rho F_i|| + sum ||rho Fg_i|| (constr) + sum ||rho T_i||_F (force);
                                        LMIs (HcE_i - I)/s_alpha >> 0, (s_beta beta I - HcE_i)/s_alpha >> 0
  * `convexHessianExprPicos` :310-357   HcE_i = s_alpha alpha H_i + calH_i(s_dP dP) + G_i' diag(s_F Fg_i) G_i
                                        + C_i' diag(s_F F_i) C_i + s_T T_i
  * `autoScaling`            :374-401
  * `check_convergence`      :403-456   un-scaling by 1/(s_alpha alpha), supplement, eigenvalue status
  * `convexify`              :36-163    Step 1 -> Step 2 (if infeasible and C given) -> Step 3 (force) or ValueError

The second-order-cone terms ||v|| <= t are written as arrow LMIs [[t, v'], [v, t I]] >> 0 and the elementwise bounds as a
linear cone, so one primal-dual path-following method (HKM direction on the matrix blocks, Mehrotra predictor-corrector)
covers the whole model.  It is an O(m^3) dense code for problems with a handful of stages."""
import numpy as np

try:
    from . import convexify_oracle as co
except ImportError:                                   # tests put oracle/ itself on sys.path
    import convexify_oracle as co

ALPHA_MIN = 1e-8                                      # convexifier.py:245


def _svec_basis(n):
    """basis of S^n in the order of a 'symmetric' PICOS variable's free entries: E_ab = e_a e_b' + e_b e_a' (a < b), e_a e_a'."""
    bas = []
    for a in range(n):
        for b in range(a, n):
            E = np.zeros((n, n)); E[a, b] = 1.0; E[b, a] = 1.0
            bas.append(E)
    return bas


class Model:
    """min c'y  s.t.  C_j + L_j y >> 0 (matrix blocks),  c_lp + L_lp y >= 0 (linear cone)."""

    def __init__(self):
        self.names = []            # (name, offset, size)
        self.m = 0
        self.blocks = []           # (L (nj*nj x m) as list of (col, matrix) pairs before assembly, C)
        self.lp_rows = []          # (row vector entries dict, constant)
        self.cost = {}

    def add_var(self, name, size):
        self.names.append((name, self.m, size)); self.m += size
        return self.m - size

    def var(self, name):
        for nm, off, size in self.names:
            if nm == name:
                return off, size
        raise KeyError(name)

    def add_lmi(self, const, terms):
        """const (nj x nj) + sum_k y[idx_k] * M_k >> 0; terms: list of (idx, matrix)."""
        self.blocks.append((np.asarray(const, dtype=np.float64), terms))

    def add_pos(self, const, terms):
        """const + sum_k coef_k y[idx_k] >= 0; terms: list of (idx, coef)."""
        self.lp_rows.append((float(const), terms))

    def assemble(self):
        blocks = []
        for const, terms in self.blocks:
            nj = const.shape[0]
            L = np.zeros((nj * nj, self.m))
            for idx, M in terms:
                L[:, idx] += np.asarray(M, dtype=np.float64).ravel()
            blocks.append((L, const))
        Llp = np.zeros((len(self.lp_rows), self.m)); clp = np.zeros(len(self.lp_rows))
        for r, (const, terms) in enumerate(self.lp_rows):
            clp[r] = const
            for idx, coef in terms:
                Llp[r, idx] += coef
        c = np.zeros(self.m)
        for idx, coef in self.cost.items():
            c[idx] += coef
        return blocks, Llp, clp, c


def set_up_model(A, B, Q, R, N, G=None, C=None, rho=1e-3, constr=True, force=False):
    """`setUpModelPicos` (convexifier.py:213-308).  A, B, Q, R, N: lists of p matrices; G: list of p (ng x n) or None; C: list of
    p ((nc_i x n) or None) or None.  Returns (Model, scaling dict, H list)."""
    p = len(A)
    nx = A[0].shape[0]; nu = B[0].shape[1]; n = nx + nu
    H = [co.build_hessian(Q[i], R[i], N[i]) for i in range(p)]
    s, sbeta = co.auto_scaling(np.stack(H))                            # :237, :374-401: alpha = dP = F = T = 1/min|eig|, beta = max/min
    sc = {'alpha': s, 'beta': sbeta, 'dP': s, 'F': s, 'T': s}
    M = Model()
    i_alpha = M.add_var('alpha', 1)                                    # :240
    i_beta = M.add_var('beta', 1)                                      # :241
    bas_x = _svec_basis(nx); d = len(bas_x)
    i_dP = [M.add_var('dP%d' % i, d) for i in range(p)]                # :242
    M.add_pos(-ALPHA_MIN, [(i_alpha, 1.0)])                            # :245  alpha > 1e-8
    i_Fg = None
    if G is not None:                                                  # :249-255
        ng = G[0].shape[0]
        i_Fg = [M.add_var('Fg%d' % i, ng) for i in range(p)]
        for i in range(p):
            for r in range(ng):
                M.add_pos(0.0, [(i_Fg[i] + r, 1.0)])
    i_F = None
    if (C is not None) and (constr is True):                           # :258-266
        i_F = []
        for i in range(p):
            if C[i] is not None:
                nc = C[i].shape[0]
                off = M.add_var('F%d' % i, nc)
                i_F.append(off)
                for r in range(nc):
                    M.add_pos(0.0, [(off + r, 1.0)])
            else:
                i_F.append(None)
    i_T = None
    bas_n = _svec_basis(n); dn = len(bas_n)
    if force:                                                          # :269-273  (T[i] > 0: every entry of the symmetric matrix)
        i_T = [M.add_var('T%d' % i, dn) for i in range(p)]
        for i in range(p):
            for r in range(dn):
                M.add_pos(0.0, [(i_T[i] + r, 1.0)])
    # objective (:276-287): beta + sum ||rho F_i|| + sum ||rho Fg_i|| + sum ||rho T_i||_F, each norm through an epigraph variable
    M.cost[i_beta] = 1.0

    def add_norm(name, entries):
        """t >= || (coef_k y[idx_k])_k ||_2 as the arrow LMI [[t, v'], [v, t I]] >> 0, and t joins the objective."""
        it = M.add_var(name, 1)
        k = len(entries)
        terms = [(it, np.eye(k + 1))]
        for q, (idx, coef) in enumerate(entries):
            E = np.zeros((k + 1, k + 1)); E[0, q + 1] = coef; E[q + 1, 0] = coef
            terms.append((idx, E))
        M.add_lmi(np.zeros((k + 1, k + 1)), terms)
        M.cost[it] = 1.0

    if constr:
        for i in range(p):
            if i_F is not None and i_F[i] is not None:
                add_norm('tF%d' % i, [(i_F[i] + r, rho) for r in range(C[i].shape[0])])
            if G is not None:
                add_norm('tG%d' % i, [(i_Fg[i] + r, rho) for r in range(G[i].shape[0])])
    if force:
        for i in range(p):
            ent = []
            for r, E in enumerate(bas_n):                              # Frobenius norm counts an off-diagonal entry twice
                ent.append((i_T[i] + r, rho * (np.sqrt(2.0) if E.sum() == 2.0 else 1.0)))
            add_norm('tT%d' % i, ent)
    # LMIs (:304-306) with HcE_i of convexHessianExprPicos (:310-357)
    I = np.eye(n)
    for i in range(p):
        terms = [(i_alpha, sc['alpha'] * H[i])]                        # :323
        V = np.hstack([A[i], B[i]])
        kn = (i + 1) % p
        for r, E in enumerate(bas_x):
            terms.append((i_dP[kn] + r, sc['dP'] * (V.T @ E @ V)))     # :339-343 with dP2 = dP[(index+1)%period]
            Ee = np.zeros((n, n)); Ee[:nx, :nx] = E
            terms.append((i_dP[i] + r, -sc['dP'] * Ee))                # -dP1 in dQ
        if G is not None:
            for r in range(G[i].shape[0]):
                g = G[i][r:r + 1, :]
                terms.append((i_Fg[i] + r, sc['F'] * (g.T @ g)))       # :346-347
        if i_F is not None and i_F[i] is not None:
            for r in range(C[i].shape[0]):
                cr = C[i][r:r + 1, :]
                terms.append((i_F[i] + r, sc['F'] * (cr.T @ cr)))      # :348-350
        if i_T is not None:
            for r, E in enumerate(bas_n):
                terms.append((i_T[i] + r, sc['T'] * E))                # :353-355
        lo = [(idx, Mx / sc['alpha']) for idx, Mx in terms]
        M.add_lmi(-I / sc['alpha'], lo)                                                   # :305
        hi = [(idx, -Mx / sc['alpha']) for idx, Mx in terms] + [(i_beta, sc['beta'] * I / sc['alpha'])]
        M.add_lmi(np.zeros((n, n)), hi)                                                   # :306
    return M, sc, H


def _sym(X):
    return 0.5 * (X + X.T)


def _max_step(X, dX):
    Li = np.linalg.inv(np.linalg.cholesky(X))
    lm = np.linalg.eigvalsh(_sym(Li @ dX @ Li.T)).min()
    return np.inf if lm >= 0 else -1.0 / lm


def solve_model(M, y0, tol=1e-9, maxit=80, verbose=False):
    """Infeasible-start primal-dual path following (HKM direction on the matrix blocks, Mehrotra predictor-corrector).
    Returns (y, status) with status 'optimal' | 'infeasible' (divergence of the complementarity gap) | 'max_iter'."""
    blocks, Llp, clp, c = M.assemble()
    J = len(blocks); m = M.m
    sizes = [Cj.shape[0] for _, Cj in blocks]
    N = sum(sizes) + len(clp)
    y = y0.copy()
    S = [np.eye(nj) for nj in sizes]; X = [np.eye(nj) / N for nj in sizes]
    for j, (L, Cj) in enumerate(blocks):                  # start from the slack itself where it is comfortably interior
        Sy = (L @ y).reshape(sizes[j], sizes[j]) + Cj
        if np.linalg.eigvalsh(_sym(Sy)).min() > 0.1:
            S[j] = _sym(Sy)
    slp = np.maximum(clp + Llp @ y, 1.0) if len(clp) else np.zeros(0)
    xlp = np.full(len(clp), 1.0 / N)
    mu0 = None
    status = 'max_iter'
    for it in range(maxit):
        Rd = [(blocks[j][0] @ y).reshape(sizes[j], sizes[j]) + blocks[j][1] - S[j] for j in range(J)]
        rlp = clp + Llp @ y - slp
        mu = (sum(np.sum(X[j] * S[j]) for j in range(J)) + xlp @ slp) / N
        rp = c - sum(blocks[j][0].T @ X[j].ravel() for j in range(J)) - Llp.T @ xlp
        pinf = np.linalg.norm(rp) / (1.0 + np.linalg.norm(c))
        dinf = np.sqrt(sum(np.sum(r * r) for r in Rd) + rlp @ rlp) / (1.0 + np.sqrt(sum(np.sum(q * q) for q in S) + slp @ slp))
        obj = c @ y
        relgap = N * mu / max(1.0, abs(obj))
        if verbose:
            print(f"it {it:2d} obj={obj:.10f} mu={mu:.3e} pinf={pinf:.2e} dinf={dinf:.2e} relgap={relgap:.2e}")
        if mu0 is None:
            mu0 = mu
        if not np.isfinite(mu) or mu > 1e6 * mu0 or np.linalg.norm(y) > 1e12:
            status = 'infeasible'
            break
        if relgap < tol and pinf < tol and dinf < tol:
            status = 'optimal'
            break
        Sinv = [np.linalg.inv(S[j]) for j in range(J)]
        Bm = np.zeros((m, m))
        for j in range(J):
            L = blocks[j][0]
            K = 0.5 * (np.kron(X[j], Sinv[j]) + np.kron(Sinv[j], X[j]))
            Bm += L.T @ K @ L
        if len(clp):
            Bm += Llp.T @ ((xlp / slp)[:, None] * Llp)
        Bm = _sym(Bm)
        reg = 0.0
        while True:
            try:
                cf = np.linalg.cholesky(Bm + reg * np.diag(np.diag(Bm)))
                break
            except np.linalg.LinAlgError:
                reg = 1e-14 if reg == 0.0 else reg * 100.0
                if reg > 1e-2:
                    return y, 'infeasible'

        def direction(sig_mu, corr=None, corr_lp=None):
            rhs = -c.copy()
            for j in range(J):
                T = sig_mu * Sinv[j] - _sym(X[j] @ Rd[j] @ Sinv[j])
                if corr is not None:
                    T = T - corr[j]
                rhs += blocks[j][0].T @ T.ravel()
            if len(clp):
                t = sig_mu / slp - xlp * rlp / slp
                if corr_lp is not None:
                    t = t - corr_lp
                rhs += Llp.T @ t
            dy = np.linalg.solve(cf.T, np.linalg.solve(cf, rhs))
            dS = [(blocks[j][0] @ dy).reshape(sizes[j], sizes[j]) + Rd[j] for j in range(J)]
            dX = []
            for j in range(J):
                T = sig_mu * Sinv[j] - X[j] - _sym(X[j] @ dS[j] @ Sinv[j])
                if corr is not None:
                    T = T - corr[j]
                dX.append(T)
            dslp = Llp @ dy + rlp
            dxlp = sig_mu / slp - xlp - xlp * dslp / slp
            if corr_lp is not None:
                dxlp = dxlp - corr_lp
            return dy, dS, dX, dslp, dxlp

        def steps(dS, dX, dslp, dxlp):
            ap = min([_max_step(X[j], dX[j]) for j in range(J)] + [np.inf])
            ad = min([_max_step(S[j], dS[j]) for j in range(J)] + [np.inf])
            if len(clp):
                neg = dxlp < 0
                if neg.any():
                    ap = min(ap, (-xlp[neg] / dxlp[neg]).min())
                neg = dslp < 0
                if neg.any():
                    ad = min(ad, (-slp[neg] / dslp[neg]).min())
            return ap, ad

        dy, dS, dX, dslp, dxlp = direction(0.0)
        ap, ad = steps(dS, dX, dslp, dxlp)
        ap = min(1.0, ap); ad = min(1.0, ad)
        mu_aff = (sum(np.sum((X[j] + ap * dX[j]) * (S[j] + ad * dS[j])) for j in range(J)) + (xlp + ap * dxlp) @ (slp + ad * dslp)) / N
        sigma = min(max((mu_aff / mu) ** 2, 1e-8), 1.0)
        corr = [_sym(dX[j] @ dS[j] @ Sinv[j]) for j in range(J)]
        corr_lp = dxlp * dslp / slp if len(clp) else None
        dy, dS, dX, dslp, dxlp = direction(sigma * mu, corr, corr_lp)
        ap, ad = steps(dS, dX, dslp, dxlp)
        mn = min(ap, ad, 1.0)
        gam = 0.9 + 0.09 * mn
        ap = min(1.0, gam * ap); ad = min(1.0, gam * ad)
        for j in range(J):
            X[j] = _sym(X[j] + ap * dX[j]); S[j] = _sym(S[j] + ad * dS[j])
        xlp = xlp + ap * dxlp; slp = slp + ad * dslp
        y = y + ad * dy
    return y, status


def solve_step(A, B, Q, R, N, G=None, C=None, rho=1e-3, constr=True, force=False, tol=1e-9, verbose=False):
    """One `setUpModelPicos` + `solveSDP` + `check_convergence` round (convexifier.py:101-108 resp. :121-127, :146-148).
    Returns dict(status, dHc, dQc, dRc, dNc, Hc, alpha, beta, kappa, dP, Fg, F, T, solver_status)."""
    M, sc, H = set_up_model(A, B, Q, R, N, G=G, C=C, rho=rho, constr=constr, force=force)
    p = len(A); nx = A[0].shape[0]; n = H[0].shape[0]
    y0 = np.zeros(M.m)
    y0[M.var('alpha')[0]] = 1.0 / sc['beta']             # s_alpha alpha H has eigenvalues in [-1, 1]
    y0[M.var('beta')[0]] = 2.0 / sc['beta']              # s_beta beta = 2
    for name, off, size in M.names:
        if name[0] == 't' and name[1] in 'FGT':
            y0[off] = 1.0
        elif name[0] in 'FT':
            y0[off:off + size] = 1.0 / sc['beta']
    y, sstat = solve_model(M, y0, tol=tol, verbose=verbose)
    alpha = y[M.var('alpha')[0]]; beta = y[M.var('beta')[0]]
    # check_convergence (:403-456)
    bas_x = _svec_basis(nx); bas_n = _svec_basis(n)
    sa = sc['alpha'] * alpha

    def smat(off, bas):
        return sum(y[off + r] * E for r, E in enumerate(bas))
    dP = [sc['dP'] * smat(M.var('dP%d' % i)[0], bas_x) / sa for i in range(p)]                           # :406-407
    Fg = [sc['F'] * y[slice(M.var('Fg%d' % i)[0], sum(M.var('Fg%d' % i)))] / sa for i in range(p)] if G is not None else None   # :409-411
    F = None
    if constr and C is not None:                                                                          # :415-421
        F = []
        for i in range(p):
            if C[i] is not None:
                off, size = M.var('F%d' % i)
                F.append(sc['F'] * y[off:off + size] / sa)
            else:
                F.append(None)
    T = [sc['T'] * smat(M.var('T%d' % i)[0], bas_n) / sa for i in range(p)] if force else None          # :425-426, :430-431
    dHc, dQc, dRc, dNc = co.convex_hessian_suppl(np.stack(A), np.stack(B), np.stack(dP), G=G, Fg=Fg, C=C if F is not None else None, F=F, T=T)
    Hc = [H[k] + dHc[k] for k in range(p)]                                                               # :435
    min_eig = min(np.min(np.linalg.eigvals(Hc[k]).real) for k in range(p))                              # :438-440
    if sstat == 'optimal' and min_eig > 0:
        status = 'Optimal'                                                                               # :442-444
    elif min_eig > 0:
        status = 'Feasible'                                                                              # :445-447
    else:
        status = 'Infeasible'                                                                            # :448-451
    kappa = max(np.linalg.cond(Hc[k]) for k in range(p)) if min_eig > 0 else np.inf
    return dict(status=status, dHc=dHc, dQc=dQc, dRc=dRc, dNc=dNc, Hc=Hc, alpha=alpha, beta=beta, kappa=kappa, dP=dP, Fg=Fg, F=F, T=T,
                solver_status=sstat, objective=None if sstat != 'optimal' else float(M.assemble()[3] @ y))


def convexify_reference(A, B, Q, R, N, G=None, C=None, opts=None):
    """`convexify` (convexifier.py:36-163): Step 1, then Step 2 if infeasible and C is given, then Step 3 if `force`, else
    ValueError.  Returns (dHc, dQc, dRc, dNc, info) -- the reference's 4-tuple plus an info dict (step taken, status)."""
    o = {'rho': 1e-3, 'solver': 'cpu', 'force': False}
    if opts:
        o.update(opts)
    H = np.stack([co.build_hessian(Q[i], R[i], N[i]) for i in range(len(A))])
    if np.min(np.linalg.eigvalsh(co.symmetrize(H))) > 0:                                                 # :82-85
        z = np.zeros_like(H[0])
        nx = A[0].shape[0]
        return z, z[:nx, :nx], z[nx:, nx:], z[:nx, nx:], dict(step=0, status='Optimal')
    res = solve_step(A, B, Q, R, N, G=G, C=None, constr=False)                                           # Step 1 (:101-108)
    step, constr = 1, False
    if res['status'] == 'Infeasible' and C is not None:                                                  # Step 2 (:116-127)
        res = solve_step(A, B, Q, R, N, G=G, C=C, rho=o['rho'], constr=True)
        step, constr = 2, True
    if res['status'] == 'Infeasible':                                                                    # :134-157
        if o['force']:
            res = solve_step(A, B, Q, R, N, G=G, C=C if constr else None, rho=o['rho'], constr=constr, force=True)
            step = 3
        else:
            raise ValueError('Convexification is not possible if the system is not optimally operated at the optimal orbit.')
    return res['dHc'], res['dQc'], res['dRc'], res['dNc'], dict(step=step, status=res['status'], kappa=res['kappa'], result=res)
