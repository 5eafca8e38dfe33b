"""CPU oracle for the TuneMPC convexify() hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the numpy restatement of the reference algorithm
(`/root/reference/tunempc/convexifier.py`) that the HIP path is checked against.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product (`tunempc_amd`) never does.

PARITY STATUS: **parity unpinned** against the reference stack.  The reference
hands its SDP to PICOS -> CVXOPT/MOSEK (`convexifier.py:29,363`), which are
un-vendored third-party packages (picos==1.2.0.post32, `setup.py:52`; Mosek 9.0.98,
`README.md:34`) that are not installed here, and the reference's own tests hold no
golden vector for this path.  What *is* pinned:
  * the problem definition (`convexifier.py:213-357`), scaling (`:374-401`),
    un-scaling and supplement reconstruction (`:403-435`, `:165-211`), the status
    rule (`:437-456`), the step/exception logic (`:36-163`) -- restated 1:1 below;
  * the literals of `examples/convex_lqr.py:40-46` and their eigenvalues / LQR gain;
  * solver-independent invariants (see `check_invariants`).
The interior-point method itself (any correct fp64 IPM solves the same SDP) is the
build's own: an infeasible-start primal-dual path-following method (HKM direction,
Mehrotra predictor-corrector) that finishes with pure centering steps on the central
path at a fixed barrier parameter, so that the returned point is a *mathematically
defined* object (the central-path point at mu_target) and CPU/GPU parity does not
depend on iteration history.

Notation: per stage k=0..p-1: A_k (nx x nx), B_k (nx x mb),
H_k = [[Q,N],[N',R]] (n x n, n = nx+mb), V_k = [A_k B_k], E = [I 0].
    calH_k(P) = V_k' P_{k+1 mod p} V_k - E' P_k E            (convexifier.py:335-343)
    M_k = s*alpha*H_k + calH_k(Pbar)                          (:325, :357; Pbar = s*dP)
    I <= M_k <= tau I,  tau = sbeta*beta,  alpha >= 1e-8      (:245, :305-306)
    min beta                                                  (:276, :287)
"""
import numpy as np
import scipy.linalg as sla
from scipy.linalg.blas import dtrsm

try:
    from . import ddnum as dn
except ImportError:                                   # tests put oracle/ itself on sys.path
    import ddnum as dn

ALPHA_MIN = 1e-8          # convexifier.py:245
STATUS_OPTIMAL, STATUS_FEASIBLE, STATUS_INFEASIBLE = 0, 1, 2
STATUS_NAMES = {0: 'Optimal', 1: 'Feasible', 2: 'Infeasible'}

# tol: complementarity tolerance relative to the optimal value, mu_target = tol * kappa (per unit of cone
# dimension; the relative duality gap on kappa is then N*tol with N = 2*p*n + 1).  2^-25 keeps the HKM Schur
# complement a factor >= 10 away from the fp64 breakdown observed at mu ~ 1e-8.
MUT_BACKOFF_MAX = 10      # mu_t back-offs per problem (hard targets)
DEFAULT_OPTS = dict(tol=2.0 ** -25, max_iter=50, center_iter=12, center_tol=1e-9)


# --------------------------------------------------------------------------- helpers
def symmetrize(S):
    """mtools.py:33-36"""
    return (S + np.swapaxes(S, -1, -2)) / 2.0


def build_hessian(Q, R, N):
    """mtools.py:38-41  [[Q,N],[N',R]]"""
    return np.vstack((np.hstack((Q, N)), np.hstack((N.T, R))))


def input_checks(arg):
    """Behaviour of preprocessing.py:157-185: all entries lists of equal length or all bare matrices (then wrapped in
    one-element lists); constant shape along the trajectory for every key but the ragged 'C'.  Same three messages."""
    kind = type(arg['A'])
    for value in arg.values():
        assert type(value) == kind, "Input arguments should be of same type!"                      # :167
    if kind is list:
        for value in arg.values():
            assert len(value) == len(arg['A']), "Input data lists should have same length!"         # :171
    else:
        for key in tuple(arg):
            arg[key] = [arg[key]]
    for key, stages in arg.items():
        if key != 'C':
            for m in stages:
                assert np.shape(m) == np.shape(stages[0]), "Data matrices should have same size along trajectory."   # :178
    return arg


def auto_scaling(H):
    """convexifier.py:374-401.  H: [p,n,n].  Returns (s, sbeta) = (1/min|eig|, max|eig|/min|eig|),
    exact zeros excluded (:387-388).  Symmetric eigensolver on real symmetric input (the
    reference calls the general `eigvals`; identical on symmetric matrices up to rounding)."""
    ev = np.abs(np.linalg.eigvalsh(H)).ravel()
    ev = ev[ev != 0.0]
    mn = min(1e10, ev.min())       # :383 initial value 1e10
    mx = max(0.0, ev.max())
    return 1.0 / mn, mx / mn


def calH(A, B, P):
    """Supplement map, convexifier.py:191-194 / :339-343.  A [p,nx,nx], B [p,nx,mb], P [p,nx,nx] -> [p,n,n]."""
    p, nx, _ = A.shape
    V = np.concatenate([A, B], axis=2)
    Pn = np.roll(P, -1, axis=0)
    out = np.swapaxes(V, 1, 2) @ Pn @ V
    out[:, :nx, :nx] -= P
    return out


def calH_adj(A, B, G):
    """Adjoint of calH:  (calH*(G))_j = V_{j-1} G_{j-1} V_{j-1}' - E G_j E'.   G [p,n,n] -> [p,nx,nx]."""
    p, nx, _ = A.shape
    V = np.concatenate([A, B], axis=2)
    W = V @ G @ np.swapaxes(V, 1, 2)
    return np.roll(W, 1, axis=0) - G[:, :nx, :nx]


def convex_hessian_suppl(A, B, P, G=None, Fg=None, C=None, F=None, T=None):
    """convexifier.py:165-211: dHc_k = sym(calH_k(P) [+ G_k' diag(Fg_k) G_k] [+ C_k' diag(F_k) C_k] [+ T_k]); slices
    dQ, dR, dN.  G [p,ng,n], Fg [p,ng]; C, F: lists of p entries ((nc_k x n) / (nc_k) or None, :198-201); T [p,n,n]."""
    nx = A.shape[1]
    Hco = calH(A, B, P).copy()
    for i in range(A.shape[0]):
        if G is not None:
            Gi = np.asarray(G[i], dtype=np.float64)
            Hco[i] = Hco[i] + Gi.T @ np.diagflat(np.asarray(Fg[i], dtype=np.float64)) @ Gi            # :196-197
        if F is not None and C is not None and C[i] is not None:
            Ci = np.asarray(C[i], dtype=np.float64)
            Hco[i] = Hco[i] + Ci.T @ np.diagflat(np.asarray(F[i], dtype=np.float64)) @ Ci             # :198-201
        if T is not None:
            Hco[i] = Hco[i] + np.asarray(T[i], dtype=np.float64)                                      # :202-203
    dH = symmetrize(Hco)
    return dH, dH[:, :nx, :nx], dH[:, nx:, nx:], dH[:, :nx, nx:]


# ------------------------------------------------------------ svec coordinates on S^nx
def _tri_idx(nx):
    ia, ib = np.triu_indices(nx)
    return ia, ib


def _svec_grad(G, ia, ib):
    """<E_ab, G> for the basis E_ab = e_a e_b' + e_b e_a' (a<b), e_a e_a' (a=b); G symmetric [.., nx, nx]."""
    w = np.where(ia == ib, 1.0, 2.0)
    return G[..., ia, ib] * w


def _smat(v, nx, ia, ib):
    P = np.zeros(v.shape[:-1] + (nx, nx))
    P[..., ia, ib] = v
    P[..., ib, ia] = v
    return P


_DUP_CACHE = {}


def _dup(nx):
    """Duplication matrix Dn (nx^2 x d, row-major vec): vec(E_ab) = Dn[:, (ab)] for the basis E_ab of _svec_grad."""
    if nx not in _DUP_CACHE:
        import scipy.sparse as sp
        ia, ib = _tri_idx(nx)
        d = len(ia)
        rows = np.concatenate([ia * nx + ib, (ib * nx + ia)[ia != ib]])
        cols = np.concatenate([np.arange(d), np.arange(d)[ia != ib]])
        Dn = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(nx * nx, d))
        _DUP_CACHE[nx] = (Dn, Dn.T.tocsr())
    return _DUP_CACHE[nx]


def _T(L, R, ia, ib):
    """T(L,R)[(ab),(cd)] = <E_ab, L E_cd R'> = [Dn' (L kron R) Dn]  (d x d); L,R: [nx,nx] or stacked [p,nx,nx].
    Note T(L,R) == T(R,L), so the HKM block 0.5*(T(Lx,Ls)+T(Ls,Lx)) is just T(Lx,Ls)."""
    if L.ndim == 3:
        return np.stack([_T(L[k], R[k], ia, ib) for k in range(L.shape[0])])
    Dn, Dt = _dup(L.shape[0])
    K = np.kron(L, R)
    return np.asarray(Dt @ (Dn.T @ K.T).T)


def _hkm_block(Lx, Ls, ia, ib):
    """the d x d HKM Schur block(s) generated by the nx x nx pair(s) (Lx, Ls):  0.5*(T(Lx,Ls)+T(Ls,Lx)) = T(Lx,Ls)."""
    return _T(Lx, Ls, ia, ib)


# ------------------------------------------------------------ block-cyclic-tridiagonal solve
class _CyclicBlockChol:
    """Cholesky of the SPD block-cyclic-tridiagonal matrix with diagonal blocks D[k] (d x d),
    coupling blocks C[k] = T[P_k, P_{k+1 mod p}] (d x d).  p>=3 uses the structured factorisation,
    p<=2 a dense one.  `shift`: relative diagonal shift added on breakdown (Cholesky-with-shift)."""

    def __init__(self, D, C):
        p, d, _ = D.shape
        self.p, self.d = p, d
        self.shift = 0.0
        sh = 0.0
        while True:
            try:
                self._factor(D, C, sh)
                break
            except np.linalg.LinAlgError:
                sh = 1e-13 if sh == 0.0 else sh * 100.0
                if sh > 1e-2:
                    raise
        self.shift = sh

    def _factor(self, D, C, sh):
        p, d = self.p, self.d
        if p <= 2:
            T = np.zeros((p * d, p * d))
            for k in range(p):
                T[k*d:(k+1)*d, k*d:(k+1)*d] += D[k]
                kn = (k + 1) % p
                if kn == k:
                    T[k*d:(k+1)*d, k*d:(k+1)*d] += C[k] + C[k].T
                else:
                    T[k*d:(k+1)*d, kn*d:(kn+1)*d] += C[k]
                    T[kn*d:(kn+1)*d, k*d:(k+1)*d] += C[k].T
            if sh:
                T = T + sh * np.diag(np.diag(T))
            self.Ld = np.linalg.cholesky(T)
            return
        Lkk = np.zeros((p, d, d)); O = np.zeros((p, d, d)); F = np.zeros((p, d, d))
        Dw = D.copy()
        if sh:
            for k in range(p):
                Dw[k] = Dw[k] + sh * np.diag(np.diag(D[k]))
        Fpre = C[p - 1].copy()                  # block [p-1, 0]
        for k in range(p - 1):
            Lkk[k] = np.linalg.cholesky(Dw[k])
            sub = C[k].T.copy()                 # block [k+1, k]
            if k == p - 2:
                sub = sub + Fpre                # fill meets the sub-diagonal
                O[k] = dtrsm(1.0, Lkk[k], sub, side=1, lower=1, trans_a=1)
                Dw[p - 1] -= O[k] @ O[k].T
            else:
                O[k] = dtrsm(1.0, Lkk[k], sub, side=1, lower=1, trans_a=1)
                F[k] = dtrsm(1.0, Lkk[k], Fpre, side=1, lower=1, trans_a=1)
                Dw[k + 1] -= O[k] @ O[k].T
                Dw[p - 1] -= F[k] @ F[k].T
                Fpre = -F[k] @ O[k].T
        Lkk[p - 1] = np.linalg.cholesky(Dw[p - 1])
        self.Lkk, self.O, self.F = Lkk, O, F

    def solve(self, R):
        """R: [p, d, nrhs] -> solution same shape."""
        p, d = self.p, self.d
        if p <= 2:
            r = R.reshape(p * d, -1)
            z = sla.solve_triangular(self.Ld, r, lower=True)
            z = sla.solve_triangular(self.Ld.T, z, lower=False)
            return z.reshape(R.shape)
        Lkk, O, F = self.Lkk, self.O, self.F
        Z = R.copy()
        # forward
        for k in range(p - 1):
            Z[k] = sla.solve_triangular(Lkk[k], Z[k], lower=True)
            Z[k + 1] -= O[k] @ Z[k]
            if k < p - 2:
                Z[p - 1] -= F[k] @ Z[k]
        Z[p - 1] = sla.solve_triangular(Lkk[p - 1], Z[p - 1], lower=True)
        # backward
        Z[p - 1] = sla.solve_triangular(Lkk[p - 1].T, Z[p - 1], lower=False)
        for k in range(p - 2, -1, -1):
            Z[k] -= O[k].T @ Z[k + 1]
            if k < p - 2:
                Z[k] -= F[k].T @ Z[p - 1]
            Z[k] = sla.solve_triangular(Lkk[k].T, Z[k], lower=False)
        return Z


# ------------------------------------------------------------ the same system in double-double (tight-accuracy mode)
DD_SWITCH = 2.0 ** -23    # tight mode: block linear algebra in double-double once mu <= DD_SWITCH * max(1, |tau|) (or after a shifted fp64 factorisation)


def _T_dd(L, R, ia, ib):
    """_T for stacked dd factors L, R [p, nx, nx] -> dd [p, d, d] (exact products of the dd entries, dd sums)."""
    a = ia[:, None]; b = ib[:, None]; c = ia[None, :]; e = ib[None, :]
    wab = (ia != ib).astype(np.float64)[:, None]; wce = (ia != ib).astype(np.float64)[None, :]
    t = L[:, a, c] * R[:, b, e]
    t = t + (L[:, a, e] * R[:, b, c]) * wce
    t = t + (L[:, b, c] * R[:, a, e]) * wab
    t = t + (L[:, b, e] * R[:, a, c]) * (wab * wce)
    return t


def _assemble_dd(X1, S1i, X2, S2i, V, nx, ia, ib):
    """D_k, C_k of the HKM Schur matrix in double-double: the Kronecker-factor images V X V', V S^-1 V', X_E V', S^-1_E V' are formed in dd
    from the fp64 X, S^-1, V (their fp64 rounding alone, eps |S^-1| ~ eps/mu against eigenvalues ~ mu, is the wall of the fp64 path)."""
    p = X1.shape[0]
    d = len(ia)
    D = dn.zeros((p, d, d)); C = dn.zeros((p, d, d))
    Vd = dn.DD(V)
    for (X, Si) in ((X1, S1i), (X2, S2i)):
        Xd = dn.DD(X); Sd = dn.DD(Si)
        VX = dn.matmul(Vd, Xd); VS = dn.matmul(Vd, Sd)                  # [p, nx, n]
        Kx = dn.matmul_nt(VX, Vd); Ks = dn.matmul_nt(VS, Vd)            # V X V'
        Fx = dn.matmul_nt(Xd[:, :nx, :], Vd); Fs = dn.matmul_nt(Sd[:, :nx, :], Vd)
        D = D + _T_dd(Xd[:, :nx, :nx], Sd[:, :nx, :nx], ia, ib)
        Tk = _T_dd(Kx, Ks, ia, ib)
        D = D + dn.DD(np.roll(Tk.hi, 1, axis=0), np.roll(Tk.lo, 1, axis=0))
        C = C - _T_dd(Fx, Fs, ia, ib)
    return D, C


class _CyclicBlockCholDD:
    """_CyclicBlockChol in double-double: same elimination order, no shift (a non-positive pivot raises LinAlgError).
    solve() takes and returns fp64 right-hand sides / solutions; the substitutions run in dd."""
    shift = 0.0

    def __init__(self, D, C):
        p, d, _ = D.shape
        self.p, self.d = p, d
        if p <= 2:
            T = dn.zeros((p * d, p * d))
            for k in range(p):
                sl = slice(k * d, (k + 1) * d)
                T[sl, sl] = T[sl, sl] + D[k]
                kn = (k + 1) % p
                sn = slice(kn * d, (kn + 1) * d)
                if kn == k:
                    T[sl, sl] = T[sl, sl] + C[k] + C[k].T
                else:
                    T[sl, sn] = T[sl, sn] + C[k]
                    T[sn, sl] = T[sn, sl] + C[k].T
            self.Ld = dn.cholesky(T)
            return
        self.Lkk = [None] * p; self.O = [None] * p; self.F = [None] * p
        Dw = [D[k].copy() for k in range(p)]
        Fpre = C[p - 1].copy()                  # block [p-1, 0]
        for k in range(p - 1):
            L = dn.cholesky(Dw[k]); self.Lkk[k] = L
            sub = C[k].T.copy()                 # block [k+1, k]
            if k == p - 2:
                sub = sub + Fpre
                self.O[k] = dn.solve_lower(L, sub.T).T                 # sub L^-T
                Dw[p - 1] = Dw[p - 1] - dn.matmul_nt(self.O[k], self.O[k])
            else:
                self.O[k] = dn.solve_lower(L, sub.T).T
                self.F[k] = dn.solve_lower(L, Fpre.T).T
                Dw[k + 1] = Dw[k + 1] - dn.matmul_nt(self.O[k], self.O[k])
                Dw[p - 1] = Dw[p - 1] - dn.matmul_nt(self.F[k], self.F[k])
                Fpre = -dn.matmul_nt(self.F[k], self.O[k])
        self.Lkk[p - 1] = dn.cholesky(Dw[p - 1])

    def solve(self, R):
        p, d = self.p, self.d
        if p <= 2:
            z = dn.solve_lower(self.Ld, R.reshape(p * d, -1))
            z = dn.solve_lower(self.Ld, z, trans=True)
            return z.to_float().reshape(R.shape)
        Lkk, O, F = self.Lkk, self.O, self.F
        Z = [dn.DD(R[k]) for k in range(p)]
        for k in range(p - 1):
            Z[k] = dn.solve_lower(Lkk[k], Z[k])
            Z[k + 1] = Z[k + 1] - dn.matmul(O[k], Z[k])
            if k < p - 2:
                Z[p - 1] = Z[p - 1] - dn.matmul(F[k], Z[k])
        Z[p - 1] = dn.solve_lower(Lkk[p - 1], Z[p - 1])
        Z[p - 1] = dn.solve_lower(Lkk[p - 1], Z[p - 1], trans=True)
        for k in range(p - 2, -1, -1):
            Z[k] = Z[k] - dn.matmul(O[k].T, Z[k + 1])
            if k < p - 2:
                Z[k] = Z[k] - dn.matmul(F[k].T, Z[p - 1])
            Z[k] = dn.solve_lower(Lkk[k], Z[k], trans=True)
        return np.stack([z.to_float() for z in Z])


POLISH_ENTER = 1e-4       # tight mode: the primal-dual centering phase hands over to the dd dual-Newton polish after a full step this small
POLISH_MAX = 6
_POLISH_W64 = False


def _inv_dd(S):
    """S [p, n, n] dd, positive definite -> S^-1 (dd) by Cholesky and two substitutions; raises LinAlgError otherwise."""
    L = dn.cholesky(S)
    out = dn.zeros(S.shape)
    eye = np.eye(S.shape[-1])
    for k in range(S.shape[0]):
        out[k] = dn.solve_lower(L[k], dn.solve_lower(L[k], eye), trans=True)
    return out


def _polish_dd(A, B, Hb, tau, alpha, P, mu, center_tol, verbose=False, GG=None, mask=None, phi=None, arrows=None, wr=0.0, cw=None):
    """Tight mode, last phase: Newton's method on the DUAL barrier problem  min tau - mu (sum logdet S1_k + logdet S2_k + log(alpha - 1e-8))
    in y = (tau, alpha, P) alone, with every stage quantity in double-double: S_r(y) is formed in dd from the fp64 y (no cancellation in
    M - I), S_r^-1 in dd, X_r := mu S_r^-1 is not an iterate any more.  The primal-dual iteration stores X and S^-1 as fp64 matrices whose
    large part (active x active, O(1) resp. 1/mu) buries the small one (~mu resp. O(1)) under an ABSOLUTE rounding error eps, and the
    directions inside the optimal face are determined by the small part: its centred point is reproducible to ~eps/mu only (1e-7 at
    mu = 2e-12, measured).  The minimiser of the barrier problem IS the central-path point at mu; two or three steps from the end of the
    primal-dual centering phase reproduce it to ~1e-12 (two runs on inputs 1e-14 apart).  Returns (tau, alpha, P, X1, X2, ok, steps, stepn[, phi]).
    GG [p, ng, n, n], mask [p, ng], phi [p, ng] (round 5): the cost-free multipliers of the equality-constraint rows (convexifier.py:249-255) join y,
    M_k gains sum_i phi_ki g_i g_i' (in dd), the barrier -mu sum log phi_ki; their border columns are formed in dd and rounded like those of tau and alpha.
    arrows (list of dict(k, idx, t), wr, cw): the norm terms of Step 2 (convexifier.py:276-283) -- epigraph variables t_e join y with cost 1 and the barrier
    -mu logdet arrow(t_e, cw o phi[idx], wr); the arrow matrix is formed and inverted in dd (t^2 - |w v|^2 cancels to ~mu on an active norm).  Returns (..., phi, arrows)."""
    p, nx, _ = A.shape
    n = Hb.shape[1]
    d = nx * (nx + 1) // 2
    ia, ib = _tri_idx(nx)
    V = np.concatenate([A, B], axis=2)
    Vd = dn.DD(V); Hd = dn.DD(Hb); I = np.eye(n)
    wsv = np.where(ia == ib, 1.0, 2.0)
    roll1 = lambda G: dn.DD(np.roll(G.hi, 1, axis=0), np.roll(G.lo, 1, axis=0))
    sym = lambda G: (G + G.T) * 0.5
    sv = lambda G: G[:, ia, ib] * wsv
    adj = lambda G: roll1(dn.matmul_nt(dn.matmul(Vd, G), Vd)) - G[:, :nx, :nx]          # calH_adj in dd
    tr = lambda G: np.trace(G.to_float(), axis1=1, axis2=2).sum()

    ng = 0 if GG is None else GG.shape[1]
    arrows = [dict(a) for a in (arrows or [])]
    if ng:
        GGd = dn.DD(GG)
        phi = np.where(mask, phi, 1.0)

    def arrow_inv(t_, phi_):
        out = []
        for a in arrows:
            m = len(a['idx'])
            if a.get('soc'):
                # second-order cone of Step 3 (the Frobenius norm of T_k): s = (t, wr c o theta), s^-1 = (t, -s_1) / det, det = t^2 - |s_1|^2 in dd
                s1 = dn.DD(cw[a['idx']] * wr) * phi_[a['k'], a['idx']]
                td = dn.DD(np.float64(t_[len(out)]))
                det = td * td
                for i in range(m):
                    det = det - s1[i] * s1[i]
                if not (t_[len(out)] > 0.0 and det.to_float() > 0.0):
                    raise np.linalg.LinAlgError('soc')
                out.append((td / det, (-s1) / det, det))             # (s^-1_0, s^-1_1, det)
                continue
            S = dn.zeros((m + 1, m + 1))
            v = dn.DD(cw[a['idx']] * wr) * phi_[a['k'], a['idx']]              # (products of two fp64 numbers: exact in dd)
            for i in range(m + 1):
                S[i, i] = dn.DD(np.float64(t_[len(out)]))
            S[0, 1:] = v; S[1:, 0] = v
            if not t_[len(out)] > 0.0:
                raise np.linalg.LinAlgError('t')
            out.append(_inv_dd(S[None])[0])
        return out

    def cones(tau_, alpha_, P_, phi_=None):
        M = Hd * alpha_ + dn.matmul(dn.matmul(Vd.T, dn.DD(np.roll(P_, -1, axis=0))), Vd)
        M[:, :nx, :nx] = M[:, :nx, :nx] - dn.DD(P_)
        if not alpha_ - ALPHA_MIN > 0.0:
            raise np.linalg.LinAlgError('alpha')
        for i in range(ng):
            if not (phi_[:, i] > 0.0).all():
                raise np.linalg.LinAlgError('phi')
            M = M + GGd[:, i] * (phi_[:, i] * mask[:, i])[:, None, None]
        return M, _inv_dd(M - I), _inv_dd((-M) + tau_ * I)

    stepn = np.inf
    X1 = X2 = None
    ok = False
    steps = 0
    tt = np.array([a['t'] for a in arrows], dtype=np.float64)

    def ret(*a):
        for e_, ar in enumerate(arrows):
            ar['t'] = float(tt[e_])
        return a + ((phi,) if ng else ()) + ((arrows,) if arrows else ())
    try:
        M, Z1, Z2 = cones(tau, alpha, P, phi)
        Za = arrow_inv(tt, phi)
    except np.linalg.LinAlgError:
        return ret(tau, alpha, P, None, None, False, 0, stepn)
    for steps in range(1, POLISH_MAX + 1):
        s0 = alpha - ALPHA_MIN
        X1 = Z1 * mu; X2 = Z2 * mu; x0 = mu / s0
        Dm = dn.zeros((p, d, d)); Cm = dn.zeros((p, d, d))
        for (X, Z) in ((X1, Z1), (X2, Z2)):
            Kx = dn.matmul_nt(dn.matmul(Vd, X), Vd); Ks = dn.matmul_nt(dn.matmul(Vd, Z), Vd)
            Fx = dn.matmul_nt(X[:, :nx, :], Vd); Fs = dn.matmul_nt(Z[:, :nx, :], Vd)
            Dm = Dm + _T_dd(X[:, :nx, :nx], Z[:, :nx, :nx], ia, ib) + roll1(_T_dd(Kx, Ks, ia, ib))
            Cm = Cm - _T_dd(Fx, Fs, ia, ib)
        try:
            chol = _CyclicBlockCholDD(Dm, Cm)
        except np.linalg.LinAlgError:
            break
        Psi = sym(dn.matmul(X2, Z2)); Ph2 = sym(dn.matmul(dn.matmul(X2, Hd), Z2)); PhiH = sym(dn.matmul(dn.matmul(X1, Hd), Z1)) + Ph2
        U = np.stack([(-sv(adj(Psi))).to_float(), sv(adj(PhiH)).to_float()], axis=2)
        Bb = np.array([[tr(Psi), -tr(Ph2)], [-tr(Ph2), np.sum((Hd * PhiH).to_float()) + x0 / s0]])
        Y = X1 - X2
        g_tau = 1.0 - tr(X2); g_alpha = -np.sum((Hd * Y).to_float()) - x0
        rhsP = sv(adj(Y)).to_float()                              # -gradient in P, rounded to fp64 AFTER the subtractions
        rb = np.array([-g_tau, -g_alpha])
        if ng:
            # one border column per multiplier (as in sdp_step1): W = Phi_k(g g') in dd, entries at P_k and P_{k+1} only
            Ug = np.zeros((p, d, p * ng)); Bfull = np.zeros((2 + p * ng, 2 + p * ng)); Bfull[:2, :2] = Bb
            Psif = Psi.to_float(); PhiHf = PhiH.to_float(); Yf = Y.to_float()
            Wf = np.zeros((p, ng, n, n))
            for i in range(ng):
                if _POLISH_W64:      # (experiment: the multiplier columns from the fp64 roundings of X_r, S_r^-1, as the HIP path forms them with k_phi_pre)
                    Wd = dn.DD(symmetrize(X1.to_float() @ GG[:, i] @ Z1.to_float()) + symmetrize(X2.to_float() @ GG[:, i] @ Z2.to_float()))
                else:
                    Wd = sym(dn.matmul(dn.matmul(X1, GGd[:, i]), Z1)) + sym(dn.matmul(dn.matmul(X2, GGd[:, i]), Z2))
                Wf[:, i] = Wd.to_float()
                cE = (-sv(Wd[:, :nx, :nx])).to_float()                               # at P_k
                cV = sv(dn.matmul_nt(dn.matmul(Vd, Wd), Vd)).to_float()              # at P_{k+1}
                for k in range(p):
                    Ug[k, :, k * ng + i] += cE[k]
                    Ug[(k + 1) % p, :, k * ng + i] += cV[k]
            Bpp = GG.reshape(p, ng, n * n) @ Wf.reshape(p, ng, n * n).transpose(0, 2, 1)
            zf = mu / phi
            g_phi = -(np.einsum('kiab,kab->ki', GG, Yf) + zf)
            for k in range(p):
                sl = slice(2 + k * ng, 2 + (k + 1) * ng)
                Bfull[sl, sl] = symmetrize(Bpp[k][None])[0] + np.diag(np.where(mask[k], zf[k] / phi[k], 1.0))
                Bfull[0, sl] = Bfull[sl, 0] = -np.einsum('iab,ab->i', GG[k], Psif[k]) * mask[k]
                Bfull[1, sl] = Bfull[sl, 1] = np.einsum('iab,ab->i', GG[k], PhiHf[k]) * mask[k]
            Ug = Ug * mask.reshape(1, 1, p * ng)
            U = np.concatenate([U, Ug], axis=2); Bb = Bfull
            if arrows:
                na = len(arrows)
                U = np.concatenate([U, np.zeros((p, d, na))], axis=2)             # the epigraph variables do not reach P
                Bf2 = np.zeros((2 + p * ng + na, 2 + p * ng + na)); Bf2[:2 + p * ng, :2 + p * ng] = Bb
                g_t = np.zeros(na)
                for e_, a in enumerate(arrows):
                    te = 2 + p * ng + e_
                    cols = 2 + a['k'] * ng + a['idx']
                    m = len(a['idx'])
                    if a.get('soc'):
                        # x = mu s^-1; Hessian of -(mu / 2) log det(s) in s: mu (2 s^-1 s^-1' - J / det); chain rule with s = (t, wr c o theta)
                        i0, i1, det = Za[e_]
                        x0d = i0 * mu; x1d = i1 * mu
                        si = np.concatenate([[i0.to_float()], np.atleast_1d(i1.to_float())])
                        Jd = np.diag(np.concatenate([[1.0], -np.ones(m)]))
                        Hs = mu * (2.0 * np.outer(si, si) - Jd / det.to_float())
                        wc_ = wr * cw[a['idx']]
                        Bf2[te, te] = Hs[0, 0]
                        Bf2[cols, te] = wc_ * Hs[0, 1:]; Bf2[te, cols] = wc_ * Hs[0, 1:]
                        Bf2[np.ix_(cols, cols)] += np.outer(wc_, wc_) * Hs[1:, 1:]
                        g_phi[a['k'], a['idx']] -= wc_ * np.atleast_1d(x1d.to_float())
                        g_t[e_] = (1.0 - x0d).to_float()
                        continue
                    Xe = (Za[e_] * mu); Xf = Xe.to_float(); Sif = Za[e_].to_float()
                    Pe = symmetrize((dn.matmul(Xe, Za[e_])).to_float()[None])[0]
                    Bf2[te, te] = np.trace(Pe)
                    for q in range(m):
                        wq = wr * cw[a['idx'][q]]
                        Eq = np.zeros((m + 1, m + 1)); Eq[0, q + 1] = wq; Eq[q + 1, 0] = wq
                        Bf2[cols[q], te] = Bf2[te, cols[q]] = 2.0 * wq * Pe[0, q + 1]
                        Fq = symmetrize((Xf @ Eq @ Sif)[None])[0]
                        Bf2[cols, cols[q]] += 2.0 * wr * cw[a['idx']] * Fq[0, 1:]
                    g_phi[a['k'], a['idx']] -= 2.0 * wr * cw[a['idx']] * Xf[0, 1:]
                    g_t[e_] = 1.0 - (Xe[np.arange(m + 1), np.arange(m + 1)]).to_float().sum()
                Bb = Bf2
            rb = np.concatenate([rb, (-g_phi * mask).ravel()] + ([-g_t] if arrows else []))
        TU = chol.solve(U)
        Sb = Bb - (np.einsum('kdi,kdj->ij', U, TU) if U.shape[2] <= 2 else U.reshape(-1, U.shape[2]).T @ TU.reshape(-1, TU.shape[2]))
        z = chol.solve(rhsP[:, :, None])[:, :, 0]
        db = np.linalg.solve(Sb, rb - np.einsum('kdi,kd->i', U, z))
        dp = z - TU @ db
        dP = _smat(dp, nx, ia, ib)
        Mf = M.to_float()
        dM = db[1] * Hb + calH(A, B, dP)
        dphi = None
        if ng:
            dphi = db[2:2 + p * ng].reshape(p, ng) * mask
            dM = dM + np.einsum('ki,kiab->kab', dphi, GG)
        dtt = db[2 + p * ng:] if arrows else None
        stepn = np.sqrt(np.sum((dM - (db[1] / alpha) * Mf) ** 2) / np.sum(Mf ** 2))
        th = 1.0
        while True:                                               # damped only if the full step leaves the cone (not seen after the centering phase)
            try:
                M, Z1, Z2 = cones(tau + th * db[0], alpha + th * db[1], P + th * dP, phi + th * dphi if ng else None)
                if arrows:
                    Za = arrow_inv(tt + th * dtt, phi + th * dphi)
                break
            except np.linalg.LinAlgError:
                th *= 0.5
                if th < 1e-3:
                    return ret(tau, alpha, P, X1.to_float(), X2.to_float(), False, steps, stepn)
        tau += th * db[0]; alpha += th * db[1]; P = P + th * dP
        if ng:
            phi = phi + th * dphi
        if arrows:
            tt = tt + th * dtt
        if verbose:
            print(f"      polish {steps}: |dy|rel={stepn:.3e} step={th:.3f}")
        if th == 1.0 and stepn < center_tol:
            ok = True
            break
    X1 = (Z1 * mu).to_float(); X2 = (Z2 * mu).to_float()
    return ret(tau, alpha, P, X1, X2, ok, steps, stepn)


# ------------------------------------------------------------ small batched helpers
def _chol_inv(S):
    """S [p,n,n] SPD -> (L, Sinv)."""
    L = np.linalg.cholesky(S)
    n = S.shape[-1]
    Li = np.linalg.solve(L, np.broadcast_to(np.eye(n), S.shape))   # L^-1
    return L, np.swapaxes(Li, 1, 2) @ Li, Li


def _max_step(Li, dX):
    """largest theta with X + theta dX >= 0, X = L L', Li = L^-1.  [p,n,n] -> scalar."""
    W = Li @ dX @ np.swapaxes(Li, 1, 2)
    lm = np.linalg.eigvalsh(symmetrize(W))[:, 0].min()
    return np.inf if lm >= 0 else -1.0 / lm


# ------------------------------------------------------------ the SDP solve (Step 1)
def _arrow(t, v, w):
    """[[t, w v'], [w v, t I]]: the LMI form of  t >= || w v ||_2  (the epigraph of one norm term of convexifier.py:276-283)."""
    m = len(v)
    S = t * np.eye(m + 1)
    S[0, 1:] = w * v; S[1:, 0] = w * v
    return S


def _arrow_start(a, m, mu):
    """t = a + delta with mu * tr(arrow(t, v)^-1) = 1 for ||w v|| = a: the eigenvalues of the arrow matrix are t - a, t + a and
    t (m - 1 times).  Bisection on delta in [mu, (m+1) mu] (60 halvings, the same loop as k_phi_init)."""
    lo, hi = mu, (m + 1.0) * mu
    for _ in range(60):
        dl = 0.5 * (lo + hi)
        f = mu * (1.0 / dl + 1.0 / (dl + 2.0 * a) + (m - 1.0) / (dl + a)) - 1.0
        if f > 0.0:
            lo = dl
        else:
            hi = dl
    return a + 0.5 * (lo + hi)


# ------------------------------------------------------------ second-order cone block (the norm term of Step 3)
# rho*||T_k||_F <= t as the Lorentz cone (t; w c o theta) in Q^{m+1} (m = n(n+1)/2 entries of T_k; PICOS hands abs() to the solver
# as a quadratic cone as well).  Jordan algebra of Q: u o v = (u'v; u0 v1 + v0 u1), identity e = (1; 0), det u = u0^2 - |u1|^2,
# central path x o s = mu e (degree 1 per cone).  Nesterov-Todd scaling W (symmetric, W x = W^-1 s = lambda), closed forms as in
# the CVXOPT cone-programming documentation: W = beta (2 v v' - J), W^-1 = (2 J v v' J - J)/beta, J = diag(1, -I).
def _soc_det(u):
    return u[0] * u[0] - u[1:] @ u[1:]


def _soc_inv(u):
    return np.concatenate([[u[0]], -u[1:]]) / _soc_det(u)


def _soc_prod(u, v):
    return np.concatenate([[u @ v], u[0] * v[1:] + v[0] * u[1:]])


def _soc_div(lam, r):
    """solve lam o u = r"""
    dl = _soc_det(lam)
    u0 = (lam[0] * r[0] - lam[1:] @ r[1:]) / dl
    u1 = (-r[0] * lam[1:] + (dl * r[1:] + (lam[1:] @ r[1:]) * lam[1:]) / lam[0]) / dl
    return np.concatenate([[u0], u1])


def _soc_scaling(s, x):
    """NT scaling of the pair (slack s, multiplier x): returns (beta, v) with W = beta (2 v v' - J)."""
    ds_, dx_ = _soc_det(s), _soc_det(x)
    sb = s / np.sqrt(ds_); xb = x / np.sqrt(dx_)
    gam = np.sqrt((1.0 + xb @ sb) / 2.0)
    wb = (sb + np.concatenate([[xb[0]], -xb[1:]])) / (2.0 * gam)
    v = wb.copy(); v[0] += 1.0
    v /= np.sqrt(2.0 * (wb[0] + 1.0))
    return (ds_ / dx_) ** 0.25, v


def _soc_W(beta, v, u, inverse=False):
    """W u or W^-1 u"""
    if inverse:
        Jv = np.concatenate([[v[0]], -v[1:]])
        return (2.0 * Jv * (Jv @ u) - np.concatenate([[u[0]], -u[1:]])) / beta
    return beta * (2.0 * v * (v @ u) - np.concatenate([[u[0]], -u[1:]]))


def _soc_W2inv(beta, v):
    """W^-2 as a dense matrix (identity plus rank two)"""
    Jv = np.concatenate([[v[0]], -v[1:]])
    return (np.eye(len(v)) + 4.0 * (v @ v) * np.outer(Jv, Jv) - 2.0 * (np.outer(Jv, v) + np.outer(v, Jv))) / (beta * beta)


def _soc_max_step(u, du):
    """largest theta with u + theta du in Q (u in int Q)"""
    a = du[0] * du[0] - du[1:] @ du[1:]
    b = u[0] * du[0] - u[1:] @ du[1:]
    c = _soc_det(u)
    # c + 2 b th + a th^2 >= 0 and u0 + th du0 >= 0: smallest positive root
    cand = [np.inf]
    if du[0] < 0:
        cand.append(-u[0] / du[0])
    if abs(a) < 1e-300:
        if b < 0:
            cand.append(-c / (2.0 * b))
    else:
        disc = b * b - a * c
        if disc >= 0:
            sq = np.sqrt(disc)
            for r in ((-b - sq) / a, (-b + sq) / a):
                if r > 0:
                    cand.append(r)
    return min(cand)


def sdp_step1(A, B, H, opts=None, verbose=False, trace=None, G=None, C=None, rho=None, force=False, cost_free=False):
    """Solve  min beta  s.t.  alpha>=1e-8, I <= M_k <= sbeta*beta*I  (convexifier.py:213-308 with
    constr=False, force=False) for one tuning problem.  Returns dict with P (= dP of
    convexifier.py:406), alpha, beta, kappa=sbeta*beta, iterations, ipm status flags.
    G [p, ng, n] (optional): equality-constraint Jacobians; M_k gains G_k' diag(phi_k) G_k with the cost-free
    multipliers phi_k = s*Fg_k >= 0 of convexifier.py:249-255 / :346-347 (they belong to Step 1 whenever G is given).
    They are handled as extra border columns of the Schur complement (each one touches only P_k and P_{k+1}).
    C (list of p entries, (nc_k x n) or None) together with rho: Step 2 (`constr=True`, convexifier.py:116-131): multipliers
    f_k = s*F_k >= 0 of the active constraints (:258-266, term :348-350) and the objective terms rho*||F_k||, rho*||Fg_k||
    (:276-283), each norm as an epigraph variable t with the arrow LMI [[t, w v'], [w v, t I]] >> 0 and w = rho*sbeta/s (the
    objective is tau = sbeta*beta plus the sum of the t).  All of them are stage-local border columns as well.
    force (with rho): Step 3 (convexifier.py:137-147): T_k symmetric with every entry > 0 (:269-273), term s_T*T_k in HcE_k
    (:352-353) and rho*||T_k||_F in the objective (:284-285).  The n(n+1)/2 free entries of T_k are further stage-local
    multipliers whose "row" is the basis matrix E_ab instead of g g'; the Frobenius norm weighs off-diagonal entries by sqrt(2).
    cost_free (with C): the OTHER reading of convexifier.py:276-283.  The reference adds the norm terms with `picos.sum(obj, abs(rho*F[i]))`;
    in PICOS 1.2.0 the second positional parameter of picos.sum was an iterator label, not a summand ([UNVERIFIED]: the
    package is not installed), in which case the objective the solver sees is beta alone and F_k, Fg_k are cost-free multipliers -- the
    rows of C_k then behave exactly like rows of G_k (ragged).  Same as rho = 0.  The paper's objective (eq. 20a) is the default."""
    o = dict(DEFAULT_OPTS)
    if opts:
        o.update(opts)
    p, nx, _ = A.shape
    n = H.shape[1]
    mb = n - nx
    d = nx * (nx + 1) // 2
    ia, ib = _tri_idx(nx)
    s, sbeta = auto_scaling(H)
    Hb = s * H
    V = np.concatenate([A, B], axis=2)
    Vt = np.swapaxes(V, 1, 2)
    I = np.eye(n)
    N = 2 * p * n + 1
    # ---- initial point (infeasible start)
    # everything O(1) from the start: alpha*Hb has eigenvalues in [-1, 1], S2 = tau*I - alpha*Hb in [1, 3]
    tau = o.get('init_tau', 2.0)
    alpha = 1.0 / sbeta
    P = np.zeros((p, nx, nx))
    S1 = np.broadcast_to(o.get('init_s', 1.0) * I, (p, n, n)).copy()
    S2 = tau * I - alpha * Hb
    X1 = np.broadcast_to(o.get('init_x', 1.0) * I / (p * n), (p, n, n)).copy()
    X2 = X1.copy()
    s0 = alpha
    x0 = o.get('init_x', 1.0) / (p * n)
    if o.get('warm') is not None:
        # experiment hook (tests/tools/warm_start_probe.py): start from a strictly feasible dual point (P, alpha, tau given in the scaled
        # problem) with the primal blocks on its central path, X = mu0 S^-1
        wm = o['warm']
        P = np.array(wm['P'], dtype=float); alpha = float(wm['alpha']); tau = float(wm['tau'])
        Mw = alpha * Hb + calH(A, B, P)
        S1 = symmetrize(Mw - I); S2 = symmetrize(tau * I - Mw)
        mu0w = float(wm.get('mu0', 1e-2))
        X1 = mu0w * np.linalg.inv(S1); X2 = mu0w * np.linalg.inv(S2)
        s0 = alpha - ALPHA_MIN; x0 = mu0w / s0
    ng = 0                                                 # rows of [G_k; C_k] per stage (padded to the longest stage, `mask` = real rows)
    ng0 = 0                                                # of which equality-constraint rows
    constr = C is not None and (rho is not None or cost_free)
    cost_free = bool(cost_free) or (constr and rho == 0.0)
    force = bool(force) and rho is not None
    nT = n * (n + 1) // 2 if force else 0                  # free entries of T_k (Step 3)
    ncs = [0] * p
    if constr:
        ncs = [0 if C[k] is None else np.atleast_2d(np.asarray(C[k], dtype=np.float64)).shape[0] for k in range(p)]
    if G is not None:
        G = np.asarray(G, dtype=np.float64)
        ng0 = G.shape[1]
    nJ = ng0 + max(ncs)                                    # rows of [G_k; C_k]
    ng = nJ + nT
    arrows = []                                            # epigraph blocks: dict(k, idx (rows of stage k), t, X)
    if ng:
        J = np.zeros((p, nJ, n)); mask = np.zeros((p, ng), dtype=bool)
        if ng0:
            J[:, :ng0] = G; mask[:, :ng0] = True
        for k in range(p):
            if ncs[k]:
                J[k, ng0:ng0 + ncs[k]] = np.atleast_2d(np.asarray(C[k], dtype=np.float64)); mask[k, ng0:ng0 + ncs[k]] = True
        GG = np.zeros((p, ng, n, n))                       # direction matrix of every multiplier: g g' per constraint row, E_ab per entry of T_k
        GG[:, :nJ] = J[:, :, :, None] * J[:, :, None, :]
        cw = np.ones(ng)                                   # weight under the norm term
        if nT:
            ta, tb = np.triu_indices(n)
            for q in range(nT):
                GG[:, nJ + q, ta[q], tb[q]] = 1.0; GG[:, nJ + q, tb[q], ta[q]] = 1.0
            cw[nJ:] = np.where(ta == tb, 1.0, np.sqrt(2.0))
            mask[:, nJ:] = True
        # slack of phi >= 0 is phi itself, z its multiplier.  Start: phi_i = min(1, 1/|g_i|^2), so that the term g_i' phi_i g_i
        # is O(1) whatever the scaling of the Jacobian rows, and z_i = x0/phi_i on the central path (two iterations fewer on
        # average than phi = 1, and no 25+-iteration stragglers with rows of norm 5)
        g2 = np.maximum(np.sqrt(np.sum(GG * GG, axis=(2, 3))), 1e-300)        # |g g'|_F = |g|^2
        phi = np.where(mask, np.minimum(1.0, 1.0 / g2), 1.0)
        z = np.where(mask, x0 / phi, 0.0)
        N = N + int(mask.sum())
        if (constr and not cost_free) or force:
            wr = rho * sbeta / s
            for k in range(p):
                blocks = []
                if constr and not cost_free:
                    blocks = ([np.arange(ng0)] if ng0 else []) + ([ng0 + np.arange(ncs[k])] if ncs[k] else [])
                if nT:
                    blocks.append(nJ + np.arange(nT))
                for bi_, idx in enumerate(blocks):
                    m = len(idx)
                    soc = bool(nT) and bi_ == len(blocks) - 1             # the block of the entries of T_k: second-order cone
                    w2 = float(np.sum(cw[idx] ** 2))               # = m for the unweighted norms of Step 2
                    # start ON the central path of the norm term: multipliers z_i = w/sqrt(m) (the gradient of w||phi|| at equal
                    # phi_i: their stationarity residual vanishes), phi_i = x0/z_i (<= 1), X = x0 S^-1 with tr X = 1 (the cost of t),
                    # i.e. S within delta ~ x0 of the cone boundary.  (phi = 1, X = x0 I, S = O(1) leaves residuals ~ 1 per norm
                    # term; with 2p terms the first Newton steps blow mu up by three orders of magnitude and the iteration diverges.)
                    ph = min(1.0, x0 * np.sqrt(w2) / wr)
                    phi[k, idx] = ph; z[k, idx] = x0 / ph
                    if soc:
                        a_ = wr * ph * np.sqrt(w2)                        # |s_1|; s = (t, w c o phi), x = x0 s^-1 with x_0 = 1 (the cost of t)
                        t0 = 0.5 * (x0 + np.sqrt(x0 * x0 + 4.0 * a_ * a_))
                        s_ = np.concatenate([[t0], wr * cw[idx] * phi[k, idx]])
                        arrows.append(dict(k=k, idx=idx, t=t0, soc=True, x=x0 * _soc_inv(s_)))
                        N = N + 1
                        continue
                    t0 = _arrow_start(wr * ph * np.sqrt(w2), m, x0)
                    arrows.append(dict(k=k, idx=idx, t=t0, soc=False, X=x0 * np.linalg.inv(_arrow(t0, cw[idx] * phi[k, idx], wr))))
                    N = N + m + 1
    mu_t = None
    tight = bool(o.get('tight', False))       # tight-accuracy mode (plain model, and Step 1 with the cost-free multipliers of G): see DD_SWITCH
    dd_on = False
    ndd = 0
    extrap_terms = None
    phase = 0
    ncent = 0
    njam = 0
    nshiftrun = 0
    nbackoff = 0
    status = 'max_iter'
    it = 0
    shift_used = 0.0
    prev_stepn = None
    stepn = np.inf
    for it in range(o['max_iter'] + o['center_iter'] * (MUT_BACKOFF_MAX + 1) + 1):
        M = alpha * Hb + calH(A, B, P)
        if ng:
            M = M + np.einsum('ki,kiab->kab', phi, GG)
        Rd1 = (M - I) - S1
        Rd2 = (tau * I - M) - S2
        rd0 = (alpha - ALPHA_MIN) - s0
        for a in arrows:
            if a['soc']:
                a['s'] = np.concatenate([[a['t']], wr * cw[a['idx']] * phi[a['k'], a['idx']]])
            else:
                a['S'] = _arrow(a['t'], cw[a['idx']] * phi[a['k'], a['idx']], wr)
        mu = (np.sum(X1 * S1) + np.sum(X2 * S2) + x0 * s0 + (np.sum(phi * z) if ng else 0.0)
              + sum((a['x'] @ a['s']) if a['soc'] else np.sum(a['X'] * a['S']) for a in arrows)) / N
        Y = X1 - X2
        r_tau = 1.0 - np.trace(X2, axis1=1, axis2=2).sum()
        r_alpha = -np.sum(Hb * Y) - x0
        r_P = -calH_adj(A, B, Y)
        r_phi2 = 0.0
        if ng:
            r_phi = -np.einsum('kiab,kab->ki', GG, Y) - z
            for a in arrows:                                   # the arrow LMI holds 2 w phi_i off the diagonal; its own variable t has cost 1
                if a['soc']:
                    r_phi[a['k'], a['idx']] -= wr * cw[a['idx']] * a['x'][1:]
                    r_phi2 += (1.0 - a['x'][0]) ** 2
                    continue
                r_phi[a['k'], a['idx']] -= 2.0 * wr * cw[a['idx']] * a['X'][0, 1:]
                r_phi2 += (1.0 - np.trace(a['X'])) ** 2
            r_phi2 += np.sum((r_phi * mask) ** 2)
        pinf = np.sqrt(r_tau ** 2 + r_alpha ** 2 + np.sum(_svec_grad(r_P, ia, ib) ** 2) + r_phi2) / 2.0
        dinf = np.sqrt(np.sum(Rd1 ** 2) + np.sum(Rd2 ** 2) + rd0 ** 2) / (1.0 + np.sqrt(np.sum(S1 ** 2) + np.sum(S2 ** 2)))
        relgap = N * mu / max(1.0, abs(tau))
        if verbose:
            print(f"it {it:2d} ph{phase} tau={tau:.10f} alpha={alpha:.4e} mu={mu:.3e} pinf={pinf:.2e} dinf={dinf:.2e} relgap={relgap:.2e}")
        if trace is not None:
            trace.append(dict(it=it, tau=tau, mu=mu, pinf=pinf, dinf=dinf, phase=phase))
        if it == 0:
            mu0 = mu
        if not (np.isfinite(mu) and np.isfinite(tau) and mu > 0) or mu > 1e6 * mu0:
            status = 'diverged'          # dual unbounded / primal infeasible: leave with the last iterate
            break
        if mu_t is None and relgap < 1e-2 and dinf < 1e-2:
            mu_t = 2.0 ** np.round(np.log2(o['tol'] * max(1.0, abs(tau))))
            if o.get('mu_target') is not None:
                # (test hook, tests/test_gpu_hard_targets.py: centre at the barrier parameter ANOTHER implementation ended at -- on hard targets the two guard their
                # factorisations differently and may back off a different number of times; the central-path point at a given mu is the same object for both)
                mu_t = float(o['mu_target'])
        # (a full Newton step removes the linear residuals: centering may start with pinf well above the final accuracy)
        # (after a shifted factorisation the directions are inexact and pinf may sit at 1e-3 ... 1e-1 while mu has arrived: the centering
        # phase -- back-off, steps of the exact factorisation one power of two up -- is the way out; the HIP path: k_ctrl_a)
        if phase == 0 and mu_t is not None and mu <= 2.0 * mu_t and dinf < 1e-6 and (pinf < 1e-3 or nshiftrun >= 1):
            phase = 1
        if phase == 0 and nshiftrun >= 2:
            # the wall met on the way down (the last two factorisations needed a shift before mu reached 2 mu_t): the path cannot be
            # followed below the current mu -- centre at the power of two above it, if the back-off budget covers that (k_ctrl_a)
            kb = max(0, int(np.ceil(np.log2(mu / mu_t))))
            if dinf < 1e-6 and nbackoff + kb <= MUT_BACKOFF_MAX:
                mu_t *= 2.0 ** kb; nbackoff += kb; phase = 1; ncent = 0; prev_stepn = None; nshiftrun = 0
            else:
                status = 'optimal_inaccurate'
                break
        if phase == 0 and it >= o['max_iter']:
            break
        L1, S1i, L1i = _chol_inv(S1)
        L2, S2i, L2i = _chol_inv(S2)
        refactor = True
        # ---- Schur complement pieces
        # nx x nx Kronecker factors per LMI block
        if refactor:
          def _blocks64():
              D_ = np.zeros((p, d, d)); C_ = np.zeros((p, d, d))
              for (X, Si) in ((X1, S1i), (X2, S2i)):
                  Kx = V @ X @ Vt; Ks = V @ Si @ Vt            # V-side (P_{k+1})
                  Fx = X[:, :nx, :] @ Vt; Fs = Si[:, :nx, :] @ Vt   # cross  (E . V')
                  D_ += _hkm_block(X[:, :nx, :nx], Si[:, :nx, :nx], ia, ib)
                  D_ += np.roll(_hkm_block(Kx, Ks, ia, ib), 1, axis=0)
                  C_ -= _hkm_block(Fx, Fs, ia, ib)
              return D_, C_
          # tight mode: the blocks, their factorisation and the substitutions in double-double once the iterate is within DD_SWITCH of the
          # boundary (cond of the Schur matrix ~ (tau/mu)^2), or as soon as the fp64 factorisation needs a shift
          if tight and not dd_on and mu <= DD_SWITCH * max(1.0, abs(tau)):
              dd_on = True
          if o.get('_assemble') is not None:            # (experiment hook of tests/tools/tight_probe.py: block assembly in extended precision)
              D, C = o['_assemble'](X1, S1i, X2, S2i, V, nx, ia, ib)
          elif not dd_on:
              D, C = _blocks64()
          # border columns (tau, alpha)
          Psi = symmetrize(X2 @ S2i)                              # Phi2(I)
          PhiH = symmetrize(X1 @ Hb @ S1i) + symmetrize(X2 @ Hb @ S2i)
          u_tau = _svec_grad(-calH_adj(A, B, Psi), ia, ib)        # [p,d]
          u_alpha = _svec_grad(calH_adj(A, B, PhiH), ia, ib)
          b_tt = np.trace(Psi, axis1=1, axis2=2).sum()
          b_ta = -np.trace(symmetrize(X2 @ Hb @ S2i), axis1=1, axis2=2).sum()
          b_aa = np.sum(Hb * PhiH) + x0 / s0
          if not dd_on:
              chol = o.get('_chol_cls', _CyclicBlockChol)(D, C)
              if tight and chol.shift > 0.0:
                  dd_on = True
          if dd_on:
              ndd += 1
              try:
                  chol = _CyclicBlockCholDD(*_assemble_dd(X1, S1i, X2, S2i, V, nx, ia, ib))
              except np.linalg.LinAlgError:
                  status = 'optimal_inaccurate'
                  break
          shift_used = max(shift_used, chol.shift)
          nshiftrun = nshiftrun + 1 if chol.shift > 0.0 else 0
          if phase == 1 and chol.shift > 0.0 and nbackoff < MUT_BACKOFF_MAX:
              # hard target (cond(T) ~ (tau/mu)^2 passes 1/eps before the default mu_t): aim for the central-path point one power of
              # two earlier AND take the step the shifted factorisation gives towards it (the HIP path: ctrl_backoff_before_rhs in k_ctrl_b, tmpc_schur.h).  Round 3:
              # the iteration used to be repeated from the same iterate -- but the Schur matrix belongs to the iterate, not to the
              # target, so ten back-offs in a row met the same singular matrix and the problem ended 'inaccurate' at 1024 mu_t.  A
              # step towards the larger mu_t moves the iterate back up the path, where the matrix is definite again: the members
              # of scripts/robustness_sweep.py at cond(H) = 1e5 that ended Feasible now end Optimal after 1-5 back-offs.
              mu_t *= 2.0; nbackoff += 1; ncent = 0; prev_stepn = None; nshiftrun = 0
              if not o.get('backoff_step', True):      # (experiment hook: the behaviour of rounds 1-2)
                  continue
          elif (phase == 1 and chol.shift > 0.0) or (nshiftrun >= 2 and (mu_t is None or nbackoff >= MUT_BACKOFF_MAX or not o.get('backoff_step', True))):
              # numerical breakdown of the Schur factorisation that no back-off is left for: keep the last iterate, strictly feasible
              # and close to the central path at ~2 mu_t, and report it as inaccurate.  (Two shifted factorisations in a row in the MAIN
              # phase with back-offs left: the step is taken, the next iteration starts centering where it stands -- see above.)
              status = 'optimal_inaccurate'
              break
          U = np.stack([u_tau, u_alpha], axis=2)                  # [p,d,2]
          Bb = np.array([[b_tt, b_ta], [b_ta, b_aa]])
          if ng:
              # one border column per multiplier phi_{k,i}: W = Phi_k(g g') lives at stage k only, so the column has
              # entries at P_k (-W[:nx,:nx]) and P_{k+1} (V W V') and nowhere else
              W = symmetrize(X1[:, None] @ GG @ S1i[:, None]) + symmetrize(X2[:, None] @ GG @ S2i[:, None])   # [p,ng,n,n]
              Ug = np.zeros((p, d, p * ng))
              for k in range(p):
                  for i in range(ng):
                      Wk = np.zeros((p, n, n)); Wk[k] = W[k, i]
                      Ug[:, :, k * ng + i] = _svec_grad(calH_adj(A, B, Wk), ia, ib)
              U = np.concatenate([U, Ug], axis=2)
              U = np.concatenate([U, np.zeros((p, d, len(arrows)))], axis=2)       # the epigraph variables do not reach P
              nb_ = 2 + p * ng + len(arrows)
              Bfull = np.zeros((nb_, nb_)); Bfull[:2, :2] = Bb
              Bpp = GG.reshape(p, ng, n * n) @ W.reshape(p, ng, n * n).transpose(0, 2, 1)         # <A_v, Phi(A_w)>  (= (g_v'X g_w)(g_w'S^-1 g_v) summed over the two LMIs for rows); as one BLAS product per stage (Step 3: ng = n(n+1)/2 'rows')
              c_tau = -np.einsum('kiab,kab->ki', GG, Psi)                                        # <g g', -Psi_k>
              c_alpha = np.einsum('kiab,kab->ki', GG, PhiH)
              for k in range(p):
                  sl = slice(2 + k * ng, 2 + (k + 1) * ng)
                  Bfull[sl, sl] = symmetrize(Bpp[k][None])[0] + np.diag(np.where(mask[k], z[k] / phi[k], 1.0))
                  Bfull[0, sl] = c_tau[k]; Bfull[sl, 0] = c_tau[k]
                  Bfull[1, sl] = c_alpha[k]; Bfull[sl, 1] = c_alpha[k]
              for e, a in enumerate(arrows):
                  te = 2 + p * ng + e
                  if a['soc']:
                      cols = 2 + a['k'] * ng + a['idx']
                      a['beta'], a['v'] = _soc_scaling(a['s'], a['x'])
                      a['lam'] = _soc_W(a['beta'], a['v'], a['x'])
                      W2 = _soc_W2inv(a['beta'], a['v'])
                      a['W2'] = W2
                      wc_ = wr * cw[a['idx']]
                      Bfull[te, te] = W2[0, 0]
                      Bfull[cols, te] = wc_ * W2[0, 1:]; Bfull[te, cols] = wc_ * W2[0, 1:]
                      Bfull[np.ix_(cols, cols)] += np.outer(wc_, wc_) * W2[1:, 1:]
                      continue
                  a['Si'] = np.linalg.inv(a['S'])
                  Pe = symmetrize(a['X'] @ a['Si'])
                  Bfull[te, te] = np.trace(Pe)
                  cols = 2 + a['k'] * ng + a['idx']
                  m = len(a['idx'])
                  for q in range(m):
                      wq = wr * cw[a['idx'][q]]
                      Eq = np.zeros((m + 1, m + 1)); Eq[0, q + 1] = wq; Eq[q + 1, 0] = wq
                      Bfull[cols[q], te] = Bfull[te, cols[q]] = 2.0 * wq * Pe[0, q + 1]
                      Fq = symmetrize(a['X'] @ Eq @ a['Si'])
                      Bfull[cols, cols[q]] += 2.0 * wr * cw[a['idx']] * Fq[0, 1:]
              Bb = Bfull
          TU = chol.solve(U)
          # (plain model: two border columns, the einsum as in rounds 1-3 -- the tight-mode tests are sensitive to its summation order; with stage-local border
          # columns, up to 2 + p n(n+1)/2 of them in Step 3, one BLAS product: 16 s of 24 at n = 34)
          Sb = Bb - (np.einsum('kdi,kdj->ij', U, TU) if U.shape[2] <= 2 else U.reshape(-1, U.shape[2]).T @ TU.reshape(-1, TU.shape[2]))

        def direction(sig_mu, corr1=None, corr2=None, corr0=0.0, corrp=None, corre=None):
            T1 = sig_mu * S1i - symmetrize(X1 @ Rd1 @ S1i)
            T2 = sig_mu * S2i - symmetrize(X2 @ Rd2 @ S2i)
            if corr1 is not None:
                T1 = T1 - corr1; T2 = T2 - corr2
            t0 = sig_mu / s0 - x0 * rd0 / s0 - corr0
            rhs_tau = np.trace(T2, axis1=1, axis2=2).sum() - 1.0
            rhs_alpha = np.sum(Hb * (T1 - T2)) + t0
            rhs_P = _svec_grad(calH_adj(A, B, T1 - T2), ia, ib)      # [p,d]
            zsol = chol.solve(rhs_P[:, :, None])[:, :, 0]
            rbv = np.array([rhs_tau, rhs_alpha])
            if ng:
                tphi = sig_mu / phi - (corrp if corrp is not None else 0.0)
                rph = np.einsum('kiab,kab->ki', GG, T1 - T2) + tphi
                rt = np.zeros(len(arrows))
                for e, a in enumerate(arrows):
                    if a['soc']:
                        a['g'] = sig_mu * _soc_inv(a['s']) - a['x'] - (corre[e] if corre is not None else 0.0)     # dx = g - W^-2 ds
                        rph[a['k'], a['idx']] += wr * cw[a['idx']] * (a['g'][1:] + a['x'][1:])
                        rt[e] = (a['g'][0] + a['x'][0]) - 1.0
                        continue
                    a['T'] = sig_mu * a['Si'] - (corre[e] if corre is not None else 0.0)
                    rph[a['k'], a['idx']] += 2.0 * wr * cw[a['idx']] * a['T'][0, 1:]
                    rt[e] = np.trace(a['T']) - 1.0
                rbv = np.concatenate([rbv, (rph * mask).ravel(), rt])
            rb = rbv - np.einsum('kdi,kd->i', U, zsol)
            db = np.linalg.solve(Sb, rb)
            dp = zsol - TU @ db
            dtau, dalpha = db[0], db[1]
            dP = _smat(dp, nx, ia, ib)
            dM = dalpha * Hb + calH(A, B, dP)
            if ng:
                dphi = db[2:2 + p * ng].reshape(p, ng)
                dM = dM + np.einsum('ki,kiab->kab', dphi, GG)
                for e, a in enumerate(arrows):
                    a['dt'] = db[2 + p * ng + e]
                    if a['soc']:
                        a['ds'] = np.concatenate([[a['dt']], wr * cw[a['idx']] * dphi[a['k'], a['idx']]])
                        a['dx'] = a['g'] - a['W2'] @ a['ds']
                        continue
                    a['dS'] = _arrow(a['dt'], cw[a['idx']] * dphi[a['k'], a['idx']], wr)
                    a['dX'] = sig_mu * a['Si'] - a['X'] - symmetrize(a['X'] @ a['dS'] @ a['Si']) - (corre[e] if corre is not None else 0.0)
            dS1 = dM + Rd1
            dS2 = dtau * I - dM + Rd2
            dX1 = sig_mu * S1i - X1 - symmetrize(X1 @ dS1 @ S1i)
            dX2 = sig_mu * S2i - X2 - symmetrize(X2 @ dS2 @ S2i)
            if corr1 is not None:
                dX1 = dX1 - corr1; dX2 = dX2 - corr2
            ds0 = dalpha + rd0
            dx0 = sig_mu / s0 - x0 - x0 * ds0 / s0 - corr0
            if ng:
                dz = (sig_mu / phi - z - z * dphi / phi - (corrp if corrp is not None else 0.0)) * mask
                return dtau, dalpha, dP, dS1, dS2, dX1, dX2, ds0, dx0, dM, dphi, dz
            return dtau, dalpha, dP, dS1, dS2, dX1, dX2, ds0, dx0, dM, None, None

        def steps(dS1, dS2, dX1, dX2, ds0, dx0, dphi=None, dz=None):
            LX1i = np.linalg.inv(np.linalg.cholesky(X1)); LX2i = np.linalg.inv(np.linalg.cholesky(X2))
            ap = min(_max_step(LX1i, dX1), _max_step(LX2i, dX2))
            ad = min(_max_step(L1i, dS1), _max_step(L2i, dS2))
            if dx0 < 0: ap = min(ap, -x0 / dx0)
            if ds0 < 0: ad = min(ad, -s0 / ds0)
            if ng:
                if (dz < 0).any(): ap = min(ap, (-z[dz < 0] / dz[dz < 0]).min())
                if (dphi < 0).any(): ad = min(ad, (-phi[dphi < 0] / dphi[dphi < 0]).min())
            for a in arrows:
                if a['soc']:
                    ap = min(ap, _soc_max_step(a['x'], a['dx'])); ad = min(ad, _soc_max_step(a['s'], a['ds']))
                    continue
                ap = min(ap, _max_step(np.linalg.inv(np.linalg.cholesky(a['X']))[None], a['dX'][None]))
                ad = min(ad, _max_step(np.linalg.inv(np.linalg.cholesky(a['S']))[None], a['dS'][None]))
            return ap, ad

        if phase == 2:
            # Taylor coefficients of the central path mu -> y(mu) at the centred point, parametrised by t: mu(t) = (1 - t) mu_t.  Order 1 is the
            # affine-scaling direction; order k >= 2 solves the same system with the right-hand side -sum_{i<k} dX_i dS_{k-i} (the Mehrotra
            # corrector is its k = 2 term).  One factorisation, one solve per order.  y(t = 1) = sum of the coefficients is the model's limit
            # point mu -> 0.  Plain model only.  MEASUREMENT ONLY: the product returns the centred point itself.
            assert not ng, 'extrap: plain model only'
            terms = [direction(0.0)]
            c1 = 0.0; c2 = 0.0; c0 = 0.0
            for k in range(2, o['extrap'] + 1):
                c1 = c1 + symmetrize(sum(terms[i][5] @ terms[k - 2 - i][3] for i in range(k - 1)) @ S1i)
                c2 = c2 + symmetrize(sum(terms[i][6] @ terms[k - 2 - i][4] for i in range(k - 1)) @ S2i)
                c0 = c0 + sum(terms[i][8] * terms[k - 2 - i][7] for i in range(k - 1)) / s0
                tot = direction(0.0, c1, c2, c0)
                terms.append(tuple(tot[j] - sum(tm[j] for tm in terms) for j in range(10)))
            extrap_terms = [(tm[0], tm[1], tm[2]) for tm in terms]          # (dtau_k, dalpha_k, dP_k)
            break
        if phase == 0:
            dtau, dalpha, dP, dS1, dS2, dX1, dX2, ds0, dx0, dM, dphi, dz = direction(0.0)
            ap, ad = steps(dS1, dS2, dX1, dX2, ds0, dx0, dphi, dz)
            ap = min(1.0, ap); ad = min(1.0, ad)
            mu_aff = (np.sum((X1 + ap * dX1) * (S1 + ad * dS1)) + np.sum((X2 + ap * dX2) * (S2 + ad * dS2))
                      + (x0 + ap * dx0) * (s0 + ad * ds0) + (np.sum((z + ap * dz) * (phi + ad * dphi)) if ng else 0.0)
                      + sum(((a['x'] + ap * a['dx']) @ (a['s'] + ad * a['ds'])) if a['soc'] else np.sum((a['X'] + ap * a['dX']) * (a['S'] + ad * a['dS']))
                            for a in arrows)) / N
            sigma = min(max((mu_aff / mu) ** o.get('sig_exp', 2), 1e-6), 1.0)     # exponent 2: ~10 % fewer iterations than Mehrotra's 3 on this SDP family (sig_exp / gam0 / gam1: experiment hooks)
            sig_mu = sigma * mu
            if mu_t is not None:
                sig_mu = max(sig_mu, mu_t)
            corr1 = symmetrize(dX1 @ dS1 @ S1i); corr2 = symmetrize(dX2 @ dS2 @ S2i)
            corr0 = dx0 * ds0 / s0
            corrp = dz * dphi / phi if ng else None
            corre = [_soc_W(a['beta'], a['v'], _soc_div(a['lam'], _soc_prod(_soc_W(a['beta'], a['v'], a['dx']), _soc_W(a['beta'], a['v'], a['ds'], inverse=True))), inverse=True)
                     if a['soc'] else symmetrize(a['dX'] @ a['dS'] @ a['Si']) for a in arrows]
            dtau, dalpha, dP, dS1, dS2, dX1, dX2, ds0, dx0, dM, dphi, dz = direction(sig_mu, corr1, corr2, corr0, corrp, corre)
            ap, ad = steps(dS1, dS2, dX1, dX2, ds0, dx0, dphi, dz)
            mn = min(ap, ad)
            gam = o.get('gam0', 0.9) + o.get('gam1', 0.09) * min(mn, 1.0)
            ap = min(1.0, gam * ap); ad = min(1.0, gam * ad)
        else:
            ncent += 1
            dtau, dalpha, dP, dS1, dS2, dX1, dX2, ds0, dx0, dM, dphi, dz = direction(mu_t)
            ap, ad = steps(dS1, dS2, dX1, dX2, ds0, dx0, dphi, dz)
            ap = min(1.0, 0.95 * ap); ad = min(1.0, 0.95 * ad)
            # first-order relative change of the output Hc_k = M_k/(s*alpha) in this step
            dMc = dM
            stepn = np.sqrt(np.sum((dMc - (dalpha / alpha) * M) ** 2) / np.sum(M ** 2))
            if verbose:
                print(f"      center |dy|rel={stepn:.3e} ap={ap:.3f} ad={ad:.3f}")
        njam = njam + 1 if (ap < 1e-6 and ad < 1e-6) else 0
        if njam >= 2:                      # step lengths collapsed twice in a row: stop with the last iterate
            status = 'optimal_inaccurate'
            break
        X1 = symmetrize(X1 + ap * dX1); X2 = symmetrize(X2 + ap * dX2)
        S1 = symmetrize(S1 + ad * dS1); S2 = symmetrize(S2 + ad * dS2)
        x0 += ap * dx0; s0 += ad * ds0
        tau += ad * dtau; alpha += ad * dalpha; P = P + ad * dP
        if ng:
            z = z + ap * dz; phi = phi + ad * dphi
        for a in arrows:
            if a['soc']:
                a['x'] = a['x'] + ap * a['dx']; a['t'] += ad * a['dt']
                continue
            a['X'] = symmetrize(a['X'] + ap * a['dX']); a['t'] += ad * a['dt']
        if phase == 1:
            # pure Newton centering on the central path at mu_t; stop on a tiny step, on stagnation
            # at the rounding floor, or on the iteration cap
            full = (ap == 1.0 and ad == 1.0)
            # extrapolated next step from the contraction r of the last two full steps: r * stepn if Newton converged
            # linearly, r^2 * stepn in its quadratic regime; r^1.5 sits between the two and stops one iteration early
            # when the convergence is already super-linear
            est = stepn * min(1.0, stepn / prev_stepn) ** 1.5 if prev_stepn is not None else stepn
            if tight and full and stepn < POLISH_ENTER:
                status = 'polish'
                break
            if full and (stepn < o['center_tol'] or est < 0.1 * o['center_tol']):
                status = 'optimal'
                if o.get('extrap', 0):        # measurement hook (tests/tools/path_sensitivity.py): Taylor model of the central path at the returned point
                    phase = 2
                    continue
                break
            if full and prev_stepn is not None and stepn > 0.5 * prev_stepn and stepn < 1e-6:
                status = 'optimal'        # rounding floor reached
                break
            if ncent >= o['center_iter']:
                if nbackoff < MUT_BACKOFF_MAX:
                    mu_t *= 2.0; nbackoff += 1; ncent = 0; prev_stepn = None
                    continue
                status = 'optimal_inaccurate'
                break
            prev_stepn = stepn if full else None
    npolish = 0
    if status == 'polish':
        if ng and arrows:
            tau, alpha, P, Xp1, Xp2, okp, npolish, stepn, phi, arrows = _polish_dd(A, B, Hb, tau, alpha, P, mu_t, o['center_tol'], verbose, GG=GG, mask=mask, phi=phi,
                                                                                   arrows=arrows, wr=wr, cw=cw)
        elif ng:
            tau, alpha, P, Xp1, Xp2, okp, npolish, stepn, phi = _polish_dd(A, B, Hb, tau, alpha, P, mu_t, o['center_tol'], verbose, GG=GG, mask=mask, phi=phi)
        else:
            tau, alpha, P, Xp1, Xp2, okp, npolish, stepn = _polish_dd(A, B, Hb, tau, alpha, P, mu_t, o['center_tol'], verbose)
        status = 'optimal' if okp else 'optimal_inaccurate'
        if Xp1 is not None:
            X1, X2 = Xp1, Xp2; s0 = alpha - ALPHA_MIN; x0 = mu_t / s0
            if ng:
                z = np.where(mask, mu_t / phi, 0.0)
    Pst = P / (s * alpha)                               # convexifier.py:406 (sP = s_alpha = s)
    out = dict(P=Pst, alpha=alpha, beta=tau / sbeta, kappa=tau, s=s, sbeta=sbeta, iters=it + 1,
               ipm_status=status, mu=mu, mu_target=mu_t, pinf=pinf, dinf=dinf, shift=shift_used, dd_iters=ndd, polish_steps=npolish, stepn=stepn, X1=X1, X2=X2, x0=x0)
    if o.get('extrap', 0) and status == 'optimal':
        out['extrap_terms'] = extrap_terms                  # scaled variables: y(t) = (tau, alpha, Pbar) + sum_k t^k terms[k-1]
        out['Pbar'] = P
    if ng0:
        out['Fg'] = phi[:, :ng0] / (s * alpha)          # convexifier.py:410 (s_F = s_alpha = s)
    if constr:
        out['F'] = [phi[k, ng0:ng0 + ncs[k]] / (s * alpha) if ncs[k] else None for k in range(p)]     # convexifier.py:415-420
    if nT:
        Tm = np.zeros((p, n, n))
        Tm[:, ta, tb] = phi[:, nJ:] / (s * alpha); Tm[:, tb, ta] = phi[:, nJ:] / (s * alpha)            # convexifier.py:422-423 (s_T = s_alpha = s)
        out['T'] = Tm
    if (constr and not cost_free) or nT:
        out['objective'] = tau / sbeta + sum(a['t'] for a in arrows) / sbeta       # beta + sum rho ||F_k|| (+ rho ||Fg_k||) (+ rho ||T_k||_F)
    return out


def check_convergence(A, B, H, P, ipm_status, G=None, Fg=None, C=None, F=None, T=None):
    """convexifier.py:403-456 (status rule :442-451)."""
    dHc, dQc, dRc, dNc = convex_hessian_suppl(A, B, P, G=G, Fg=Fg, C=C, F=F, T=T)
    Hc = H + dHc
    ev = np.linalg.eigvalsh(Hc)
    min_eig = ev.min(); max_cond = (ev[:, -1] / ev[:, 0]).max() if min_eig > 0 else np.inf
    if min_eig > 0.0:
        st = STATUS_OPTIMAL if ipm_status == 'optimal' else STATUS_FEASIBLE
    else:
        st = STATUS_INFEASIBLE
    return st, dHc, dQc, dRc, dNc, min_eig, max_cond


def convexify_arrays(A, B, H, opts=None, verbose=False, G=None, C=None, rho=1e-3, force=False):
    """Array-level restatement of convexifier.convexify, Steps 1 to 3 (optional equality-constraint term G [p,ng,n]; C: list of
    p active-constraint Jacobians (nc_k x n) or None, used by Step 2 when Step 1 is infeasible, convexifier.py:116-131; force:
    Step 3 when the problem is still infeasible, :137-147):
    A [p,nx,nx], B [p,nx,mb], H [p,n,n] -> dict(status, step, dHc, Hc, P, alpha, beta, kappa, iters, early_exit[, Fg][, F][, T])."""
    A = np.asarray(A, float); B = np.asarray(B, float); H = symmetrize(np.asarray(H, float))
    p, nx, _ = A.shape
    n = H.shape[1]
    # already convex?  convexifier.py:82-85
    if np.linalg.eigvalsh(H)[:, 0].min() > 0:
        return dict(status=STATUS_OPTIMAL, early_exit=True, dHc=np.zeros_like(H), Hc=H.copy(),
                    P=np.zeros((p, nx, nx)), alpha=1.0, beta=0.0, kappa=0.0, iters=0)
    r = sdp_step1(A, B, H, opts, verbose, G=G)
    st, dHc, dQc, dRc, dNc, min_eig, max_cond = check_convergence(A, B, H, r['P'], r['ipm_status'], G=G, Fg=r.get('Fg'))
    r.update(status=st, step=1, early_exit=False, dHc=dHc, Hc=H + dHc, min_eig=min_eig, max_cond=max_cond)
    if st == STATUS_INFEASIBLE and C is not None:          # Step 2 (eta_F = 1): convexifier.py:116-131
        it1 = r['iters']
        r = sdp_step1(A, B, H, opts, verbose, G=G, C=C, rho=rho)
        st, dHc, dQc, dRc, dNc, min_eig, max_cond = check_convergence(A, B, H, r['P'], r['ipm_status'], G=G, Fg=r.get('Fg'), C=C, F=r['F'])
        r.update(status=st, step=2, early_exit=False, dHc=dHc, Hc=H + dHc, min_eig=min_eig, max_cond=max_cond, iters_step1=it1)
    if st == STATUS_INFEASIBLE and force:                  # Step 3 (eta_T = 1): convexifier.py:137-147, constr as left by the steps before
        constr = r['step'] == 2
        r = sdp_step1(A, B, H, opts, verbose, G=G, C=C if constr else None, rho=rho, force=True)
        st, dHc, dQc, dRc, dNc, min_eig, max_cond = check_convergence(A, B, H, r['P'], r['ipm_status'], G=G, Fg=r.get('Fg'),
                                                                      C=C if constr else None, F=r.get('F'), T=r['T'])
        r.update(status=st, step=3, early_exit=False, dHc=dHc, Hc=H + dHc, min_eig=min_eig, max_cond=max_cond)
    return r


def convexify(A, B, Q, R, N, G=None, C=None, opts=None):
    """Drop-in restatement of convexifier.convexify (convexifier.py:36-163), Steps 1 to 3."""
    arg = dict(A=A, B=B, Q=Q, R=R, N=N)
    if C is not None:
        arg['C'] = C
    if G is not None:
        arg['G'] = G
    arg = input_checks(arg)
    Gs = np.stack([np.atleast_2d(np.asarray(g, float)) for g in arg['G']]) if 'G' in arg else None
    nx = np.shape(arg['A'][0])[0]; nu = np.shape(arg['B'][0])[1]
    Hs = np.stack([build_hessian(np.asarray(q, float), np.asarray(r, float), np.asarray(nn, float))
                   for q, r, nn in zip(arg['Q'], arg['R'], arg['N'])])
    As = np.stack([np.asarray(a, float) for a in arg['A']]); Bs = np.stack([np.asarray(b, float) for b in arg['B']])
    Cs = [None if c is None else np.atleast_2d(np.asarray(c, float)) for c in arg['C']] if 'C' in arg else None
    rho = (opts or {}).get('rho', 1e-3)
    sopts = {k: v for k, v in (opts or {}).items() if k not in ('rho', 'solver', 'force')} or None
    res = convexify_arrays(As, Bs, Hs, sopts, G=Gs, C=Cs, rho=rho, force=bool((opts or {}).get('force', False)))
    if res['early_exit']:
        return np.zeros((nx + nu, nx + nu)), np.zeros((nx, nx)), np.zeros((nu, nu)), np.zeros((nx, nu))   # :85
    if res['status'] == STATUS_INFEASIBLE and not (opts or {}).get('force', False):
        raise ValueError('Convexification is not possible if the system is not optimally operated at the optimal orbit.')
    dH = res['dHc']
    return ([dH[k] for k in range(len(dH))], [dH[k][:nx, :nx] for k in range(len(dH))],
            [dH[k][nx:, nx:] for k in range(len(dH))], [dH[k][:nx, nx:] for k in range(len(dH))])


# ------------------------------------------------------------ synthetic inputs
def gen_problem(seed, p, nx, mb, sigP=1.0, identity=False):
    """'random SPD-perturbed Hessian' generator: H_k = Hhat_k - calH_k(Phat), strictly feasible by construction."""
    rng = np.random.default_rng(seed)
    n = nx + mb
    A = np.zeros((p, nx, nx)); B = np.zeros((p, nx, mb)); Phat = np.zeros((p, nx, nx)); Hhat = np.zeros((p, n, n))
    for k in range(p):
        a = rng.standard_normal((nx, nx)) / np.sqrt(nx)
        rho = np.max(np.abs(np.linalg.eigvals(a)))
        A[k] = a * (0.9 / rho)
        B[k] = rng.standard_normal((nx, mb)) / np.sqrt(nx)
        W, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.ones(n) if identity else 10.0 ** rng.uniform(0, 1, n)
        Hhat[k] = (W * lam) @ W.T
        pk = rng.standard_normal((nx, nx)); Phat[k] = sigP * (pk + pk.T) / 2
    H = symmetrize(Hhat - calH(A, B, Phat))
    return A, B, H, Phat, Hhat


def gen_batch(base_seed, nb, p, nx, mb, **kw):
    out = [gen_problem(base_seed + b, p, nx, mb, **kw) for b in range(nb)]
    return tuple(np.stack([o[i] for o in out]) for i in range(3))


def check_invariants(A, B, H, res, tol_struct=1e-10):
    """Solver-independent checks.  Returns dict of measured quantities."""
    Hc = res['Hc']; P = res['P']
    ev = np.linalg.eigvalsh(Hc)
    struct = np.linalg.norm(Hc - H - symmetrize(calH(A, B, P))) / max(1.0, np.linalg.norm(Hc))
    out = dict(min_eig=ev.min(), max_cond=(ev[:, -1] / ev[:, 0]).max(), struct_err=struct)
    if not res.get('early_exit'):
        out['cond_bound'] = res['kappa']
    return out
