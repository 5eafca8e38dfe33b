"""Double-double ("dd", ~106-bit significand) arithmetic on numpy arrays  --  TEST INFRASTRUCTURE ONLY (part of the CPU oracle).

Used by the tight-accuracy mode of `convexify_oracle.sdp_step1` (opts['tight']): the Kronecker-factor images, the assembly of the
block-cyclic-tridiagonal Schur matrix, its Cholesky factorisation and the substitutions run in dd, everything else stays fp64.  The
HKM Schur matrix has eigenvalues from mu (directions inside the optimal face) to 1/mu (active x active), i.e. cond ~ 1/mu^2: fp64
normal equations lose the small end at mu ~ sqrt(eps) = 1e-8; dd (eps ~ 1e-32) carries it to mu ~ 1e-13 (tests/tools/tight_probe.py
measured the same with 40-digit mpmath arithmetic).

A value is the unevaluated sum hi + lo with |lo| <= ulp(hi)/2.  numpy has no fused multiply-add, so products are split by
Veltkamp / Dekker (exact as long as nothing overflows 2^996).  Algorithms: Dekker (1971), Knuth TAOCP 4.2.2, Hida-Li-Bailey (QD)."""
import numpy as np

_SPLIT = 134217729.0          # 2^27 + 1


def two_sum(a, b):
    s = a + b
    v = s - a
    return s, (a - (s - v)) + (b - v)


def quick_two_sum(a, b):      # |a| >= |b|
    s = a + b
    return s, b - (s - a)


def _split(a):
    t = _SPLIT * a
    hi = t - (t - a)
    return hi, a - hi


def two_prod(a, b):
    p = a * b
    ah, al = _split(a)
    bh, bl = _split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


class DD:
    """array of double-double numbers; supports + - * / (with DD, ndarray or float), unary -, indexing / slicing / assignment, .T,
    reshape-free helpers matmul_nt, sqrt.  Shapes broadcast like numpy."""
    __slots__ = ('hi', 'lo')
    __array_priority__ = 100

    def __init__(self, hi, lo=None):
        self.hi = np.asarray(hi, dtype=np.float64)
        self.lo = np.zeros_like(self.hi) if lo is None else np.asarray(lo, dtype=np.float64)

    # ---- structure
    @property
    def shape(self):
        return self.hi.shape

    @property
    def T(self):
        return DD(np.swapaxes(self.hi, -1, -2), np.swapaxes(self.lo, -1, -2))

    def copy(self):
        return DD(self.hi.copy(), self.lo.copy())

    def __getitem__(self, idx):
        return DD(self.hi[idx], self.lo[idx])

    def __setitem__(self, idx, v):
        v = _dd(v)
        self.hi[idx] = v.hi
        self.lo[idx] = v.lo

    def to_float(self):
        return self.hi + self.lo

    # ---- arithmetic
    def __neg__(self):
        return DD(-self.hi, -self.lo)

    def __add__(self, o):
        o = _dd(o)
        s, e = two_sum(self.hi, o.hi)
        t, f = two_sum(self.lo, o.lo)
        e = e + t
        s, e = quick_two_sum(s, e)
        e = e + f
        return DD(*quick_two_sum(s, e))

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-_dd(o))

    def __rsub__(self, o):
        return _dd(o) + (-self)

    def __mul__(self, o):
        o = _dd(o)
        p, e = two_prod(self.hi, o.hi)
        e = e + (self.hi * o.lo + self.lo * o.hi)
        return DD(*quick_two_sum(p, e))

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = _dd(o)
        q1 = self.hi / o.hi
        r = self - o * q1
        q2 = r.hi / o.hi
        r = r - o * q2
        q3 = r.hi / o.hi
        q = DD(*quick_two_sum(q1, q2))
        return q + q3

    def sqrt(self):
        """Karp's trick: x = 1/sqrt(a.hi); sqrt(a) ~ a.hi x + (a - (a.hi x)^2) x / 2"""
        x = 1.0 / np.sqrt(self.hi)
        ax = self.hi * x
        err = (self - DD(*two_prod(ax, ax))).hi
        return DD(*quick_two_sum(ax, err * (x * 0.5)))


def _dd(v):
    return v if isinstance(v, DD) else DD(np.asarray(v, dtype=np.float64))


def zeros(shape):
    return DD(np.zeros(shape), np.zeros(shape))


def matmul_nt(A, B):
    """A [..., m, k] times B [..., n, k]' -> [..., m, n], dd accumulation over k (K numpy steps of an outer product each)."""
    A = _dd(A); B = _dd(B)
    K = A.shape[-1]
    out = zeros(np.broadcast_shapes(A.shape[:-2], B.shape[:-2]) + (A.shape[-2], B.shape[-2]))
    for k in range(K):
        out = out + A[..., :, k:k + 1] * B[..., None, :, k]
    return out


def matmul(A, B):
    return matmul_nt(A, _dd(B).T)


def cholesky(A):
    """lower Cholesky factor of the dd matrices A [..., d, d] (right-looking); raises np.linalg.LinAlgError on a non-positive pivot."""
    W = _dd(A).copy()
    d = W.shape[-1]
    for j in range(d):
        piv = W[..., j, j]
        if not np.all(piv.hi > 0.0):
            raise np.linalg.LinAlgError('dd cholesky: non-positive pivot at column %d' % j)
        r = piv.sqrt()
        W[..., j, j] = r
        if j + 1 < d:
            col = W[..., j + 1:, j] / r[..., None]
            W[..., j + 1:, j] = col
            W[..., j + 1:, j + 1:] = W[..., j + 1:, j + 1:] - col[..., :, None] * col[..., None, :]
    # zero the strict upper triangle
    iu = np.triu_indices(d, 1)
    W.hi[..., iu[0], iu[1]] = 0.0
    W.lo[..., iu[0], iu[1]] = 0.0
    return W


def solve_lower(L, Bm, trans=False):
    """X with L X = B (trans=False) or L' X = B (trans=True); L [d, d] lower dd, B [d, m] (DD or float)."""
    X = _dd(Bm).copy()
    d = L.shape[-1]
    if not trans:
        for j in range(d):
            X[j, :] = X[j, :] / L[j, j]
            if j + 1 < d:
                X[j + 1:, :] = X[j + 1:, :] - L[j + 1:, j][:, None] * X[j, :][None, :]
    else:
        for j in range(d - 1, -1, -1):
            X[j, :] = X[j, :] / L[j, j]
            if j > 0:
                X[:j, :] = X[:j, :] - L[j, :j][:, None] * X[j, :][None, :]
    return X
