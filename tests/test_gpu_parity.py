"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libtunempc_hip.so.
Floating point: H/Hc must match the CPU oracle within 1e-8 relative Frobenius norm (BASELINE.json north_star);
the tolerance is written at each assert."""
import os

import numpy as np
import pytest
import scipy.linalg as sla
import torch  # noqa: F401  (before the HIP library is loaded: torch ships its own HIP runtime and fails to find the GPU when it initialises second)

pytestmark = pytest.mark.gpu

import convexify_oracle as co  # noqa: E402  (tests are the only place the product meets the oracle)

PARITY = 1e-8       # relative Frobenius norm, BASELINE.json: "H/q within 1e-8 of reference"


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope='module')
def hc():
    from tunempc_amd._lib import HipConvexifier
    cache = {}

    def get(p, nx, mb, **kw):
        key = (p, nx, mb, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = HipConvexifier(p, nx, mb, **kw)
        return cache[key]
    yield get
    for h in cache.values():
        h.close()


def test_library_loaded_in_tree():
    from tunempc_amd._lib import load_library, library_path
    lib = load_library()
    assert os.path.dirname(library_path()).endswith(os.path.join('tunempc_amd', 'lib'))
    assert lib.tmpc_device_count() >= 1


# ----------------------------------------------------------------------------- building blocks
@pytest.mark.parametrize('M,N,K', [(64, 64, 16), (128, 64, 48), (304, 48, 256), (304, 304, 304), (16, 16, 16), (48, 112, 80), (320, 128, 64)])
@pytest.mark.parametrize('flags', [0, 1])
def test_mfma_gemm_nt(hc, M, N, K, flags):
    """Register-staged v_mfma_f64_4x4x4 tile GEMM (and its scalar-FMA twin) vs numpy; asymmetric operands catch transposes."""
    h = hc(2, 3, 1)
    h.set_options(flags=flags)
    rng = np.random.default_rng(M * 1000 + N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); C0 = rng.standard_normal((M, N))
    for mode, ref in [(0, C0 - A @ B.T), (1, A @ B.T), (2, -A @ B.T)]:
        out = h.debug_gemm_nt(C0, A, B, mode)
        assert rel(out, ref) < 1e-14
    h.set_options(flags=0)


@pytest.mark.parametrize('M,N,K', [(64, 64, 16), (128, 64, 48), (304, 48, 256), (304, 304, 304), (16, 16, 16), (48, 112, 80), (320, 128, 64), (64, 64, 32)])
@pytest.mark.parametrize('shape', [16, 32])
def test_dma_tile_gemm(hc, M, N, K, shape):
    """The LDS-DMA tile core of the batched factorisation kernels (tmpc_gemm_dma.h: source-side swizzle, b128 fragment reads, K pairs
    split over two MFMAs, 16-column wave strips) vs numpy, every mode; shape 32: K as two operand pairs in one stream."""
    if shape == 32 and K % 32:
        pytest.skip('two equal operand pairs need K % 32 == 0')
    h = hc(2, 3, 1)
    rng = np.random.default_rng(M * 1000 + N + K + shape)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); C0 = rng.standard_normal((M, N))
    for mode, ref in [(0, C0 - A @ B.T), (1, A @ B.T), (2, -A @ B.T)]:
        out = h.debug_gemm_nt(C0, A, B, mode + shape)
        assert rel(out, ref) < 1e-14


@pytest.mark.parametrize('n,shape,G', [(304, 0, 64), (304, 16, 16), (320, 16, 16), (496, 16, 16), (304, 32, 16)])
def test_gemm_nt_lower(hc, n, shape, G):
    """Lower-only symmetric update C -= A A': the lower triangle is exact; everything above the block diagonal of G x G blocks is untouched
    (register-staged core: 64 x 64 tiles above the diagonal are skipped; LDS-DMA core: every 16 x 16 block above it)."""
    h = hc(2, 3, 1)
    rng = np.random.default_rng(n + shape)
    A = rng.standard_normal((n, n if shape != 32 else 320)); C0 = rng.standard_normal((n, n))
    out = h.debug_gemm_nt(C0, A, A, 0 + shape, lower=True)
    ref = C0 - A @ A.T
    low = np.tril(np.ones((n, n), bool))
    assert np.abs(out - ref)[low].max() < 1e-12 * n
    up = np.kron(np.triu(np.ones(((n + G - 1) // G,) * 2), 1), np.ones((G, G)))[:n, :n] > 0
    assert np.abs(out - C0)[up].max() == 0.0


def _spd_cyclic(rng, p, d):
    E = rng.standard_normal((p, 2 * d, d)) / np.sqrt(2 * d); F = rng.standard_normal((p, 2 * d, d)) / np.sqrt(2 * d)
    D = np.stack([np.eye(d) for _ in range(p)]); Cc = np.zeros((p, d, d))
    for k in range(p):
        D[k] += E[k].T @ E[k]; D[(k + 1) % p] += F[k].T @ F[k]; Cc[k] = E[k].T @ F[k]
    T = np.zeros((p * d, p * d))
    for k in range(p):
        T[k*d:(k+1)*d, k*d:(k+1)*d] += D[k]
        kn = (k + 1) % p
        if kn == k:
            T[k*d:(k+1)*d, k*d:(k+1)*d] += Cc[k] + Cc[k].T
        else:
            T[k*d:(k+1)*d, kn*d:(kn+1)*d] += Cc[k]; T[kn*d:(kn+1)*d, k*d:(k+1)*d] += Cc[k].T
    return D, Cc, T


@pytest.mark.parametrize('p,d', [(1, 6), (2, 10), (3, 10), (5, 21), (4, 78), (3, 136), (3, 300), (7, 45), (2, 3), (6, 20), (8, 33), (9, 16), (13, 10),
                                 (16, 70), (30, 10), (64, 24), (200, 5), (3, 330), (4, 410), (2, 500), (3, 320), (5, 250)])
def test_block_cyclic_cholesky_solve(hc, p, d):
    """Cyclic-reduction block factorisation + solves (tmpc_cr.h) vs a dense numpy solve of the same SPD block-cyclic-tridiagonal system
    (d <= 320: the LDS-DMA solve kernel with up to five column tiles in registers; wider blocks: the left-looking strips)."""
    h = hc(2, 3, 1)
    rng = np.random.default_rng(p * 100 + d)
    D, Cc, T = _spd_cyclic(rng, p, d)
    rhs = rng.standard_normal((p, d))
    x, nshift = h.debug_block_solve(D, Cc, rhs)
    xref = np.linalg.solve(T, rhs.ravel()).reshape(p, d)
    assert nshift == 0
    assert rel(x, xref) < 1e-12


@pytest.mark.parametrize('p,d,dead', [(3, 10, [(0, 3)]), (4, 78, [(1, 0), (1, 77), (3, 40)]), (5, 40, [(2, 17), (2, 18)]), (3, 300, [(0, 150), (2, 299)])])
def test_block_cholesky_frozen_pivots(hc, p, d, dead):
    """Cholesky-with-shift of the 16 x 16 pivot blocks (wave_potrf16, the slow path of its column step): a variable whose row and column of
    the matrix vanish has a zero pivot and a zero pivot reference; the rule of round 1 freezes it (pivot := 1e20), counts it, and leaves
    the rest of the solve untouched -- the other unknowns equal the dense solve of the system without it, its own comes out ~ 0."""
    h = hc(2, 3, 1)
    rng = np.random.default_rng(7 * p + d)
    D, Cc, T = _spd_cyclic(rng, p, d)
    for (k, i) in dead:
        D[k][i, :] = 0.0; D[k][:, i] = 0.0
        Cc[k][i, :] = 0.0; Cc[(k - 1) % p][:, i] = 0.0
    rhs = rng.standard_normal((p, d))
    x, nshift = h.debug_block_solve(D, Cc, rhs)
    keep = np.ones(p * d, bool)
    for (k, i) in dead:
        keep[k * d + i] = False
    T2 = np.zeros((p * d, p * d))
    for k in range(p):
        kn = (k + 1) % p
        T2[k*d:(k+1)*d, k*d:(k+1)*d] += D[k]
        T2[k*d:(k+1)*d, kn*d:(kn+1)*d] += Cc[k]; T2[kn*d:(kn+1)*d, k*d:(k+1)*d] += Cc[k].T
    xref = np.zeros(p * d)
    xref[keep] = np.linalg.solve(T2[np.ix_(keep, keep)], rhs.ravel()[keep])
    assert nshift == len(dead)
    assert np.abs(x.ravel()[~keep]).max() < 1e-15
    assert rel(x.ravel()[keep], xref[keep]) < 1e-11


@pytest.mark.parametrize('n', [1, 2, 4, 5, 15, 30, 32])
def test_jacobi_eig_scan(hc, n):
    """batched symmetric eigenvalue extremes (pre-check :82, autoScaling :374-401, status :438-440) vs LAPACK."""
    nx = max(n - 1, 1); mb = n - nx
    h = hc(3, nx, mb)
    rng = np.random.default_rng(n)
    H = rng.standard_normal((4, 3, n, n)); H = H + H.transpose(0, 1, 3, 2)
    H[0, 0] = np.diag(np.arange(n, dtype=float))            # exact zero eigenvalue is excluded from min|eig|
    out = h.eig_scan(H)
    ev = np.linalg.eigvalsh(H)
    aev = np.abs(ev)
    amin = np.where(aev == 0, np.inf, aev).min(-1)
    if n == 1:
        amin[0, 0] = 1e300
    ref = np.stack([ev[..., 0], ev[..., -1], amin, aev.max(-1)], -1)
    scale = np.abs(ev).max()
    assert np.abs(out[..., :2] - ref[..., :2]).max() < 1e-13 * scale
    fin = np.isfinite(ref[..., 2]) & (ref[..., 2] < 1e299)
    assert np.abs(out[..., 2][fin] - ref[..., 2][fin]).max() < 1e-13 * scale
    assert np.abs(out[..., 3] - ref[..., 3]).max() < 1e-13 * scale


@pytest.mark.parametrize('n', [1, 2, 3, 5, 16, 31, 32])
def test_tridiagonal_min_eig(hc, n):
    """Householder + Sturm multisection (the step-length primitive) vs LAPACK, incl. clustered and degenerate spectra."""
    h = hc(2, 3, 1)
    rng = np.random.default_rng(100 + n)
    W = rng.standard_normal((40, n, n)); W = W + W.transpose(0, 2, 1)
    W[0] = np.eye(n) * 3.0                                   # already diagonal, all eigenvalues equal
    W[1] = np.diag(np.linspace(-2, 5, n)) if n > 1 else W[1]
    if n > 2:
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        W[2] = (q * np.r_[-1.0, -1.0 + 1e-9, np.linspace(0, 1, n - 2)]) @ q.T      # nearly double smallest eigenvalue
        W[3] = 1e-8 * W[3]                                                           # tiny scale
        W[4] = 1e6 * W[4]
    out = h.debug_min_eig(W)
    ref = np.linalg.eigvalsh((W + W.transpose(0, 2, 1)) / 2)[:, 0]
    scale = np.abs(np.linalg.eigvalsh((W + W.transpose(0, 2, 1)) / 2)).max(-1)
    assert (np.abs(out - ref) <= 1e-12 * np.maximum(scale, 1e-300)).all(), np.abs(out - ref).max()


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 6, 7, 8])
def test_lane_min_eig(hc, n):
    """Round 5: the step-length primitive of the small shapes -- one THREAD per matrix, Householder tridiagonalisation + Laguerre's iteration in registers
    (lane_min_eig8) -- vs LAPACK: random, multiples of the identity (an n-fold root), diagonal, (nearly) double smallest eigenvalues, scales 1e-8 ... 1e8, zero."""
    h = hc(2, 3, 1)
    rng = np.random.default_rng(500 + n)
    W = rng.standard_normal((300, n, n)); W = W + W.transpose(0, 2, 1)
    W[0] = np.eye(n) * 3.0; W[1] = -np.eye(n) * 0.25; W[2] = 0.0
    W[3] = np.diag(np.linspace(-2, 5, n)) if n > 1 else W[3]
    for i in range(4, 60):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.sort(rng.standard_normal(n))
        if n > 1 and i % 3 == 0:
            lam[1] = lam[0]                                   # double smallest eigenvalue (Laguerre converges linearly there)
        if n > 1 and i % 3 == 1:
            lam[1] = lam[0] + 1e-9
        if i % 3 == 2:
            lam = 10.0 ** rng.uniform(-8, 8, n) * rng.choice([-1, 1], n)
        W[i] = (q * lam) @ q.T
    W[60] *= 1e-8; W[61] *= 1e6
    out = h.debug_min_eig(W, lane=True)
    sym = (W + W.transpose(0, 2, 1)) / 2
    ev = np.linalg.eigvalsh(sym)
    assert (np.abs(out - ev[:, 0]) <= 1e-13 * np.maximum(np.abs(ev).max(-1), 1e-300)).all(), np.abs(out - ev[:, 0]).max()


@pytest.mark.parametrize('p,nx,mb', [(1, 3, 1), (4, 4, 2), (3, 24, 8)])
def test_supplement(hc, p, nx, mb):
    """convexHessianSuppl (convexifier.py:165-211) on the GPU vs numpy."""
    h = hc(p, nx, mb)
    rng = np.random.default_rng(7)
    A = rng.standard_normal((2, p, nx, nx)); B = rng.standard_normal((2, p, nx, mb))
    P = rng.standard_normal((2, p, nx, nx)); P = P + P.transpose(0, 1, 3, 2)
    out = h.supplement_batch(A, B, P)
    for b in range(2):
        ref, _, _, _ = co.convex_hessian_suppl(A[b], B[b], P[b])
        assert rel(out[b], ref) < 1e-14


# ----------------------------------------------------------------------------- full path vs oracle
@pytest.mark.parametrize('name', ['c1_convex_lqr', 'c2_unicycle_shape', 'c3_evaporation_shape', 'mid_n16', 'awe_shape_n15',
                                  'identity_family'])
def test_golden_vectors(hc, golden_dir, name):
    """HIP path vs the committed golden vectors (inputs + oracle outputs)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    A, B, H = g['A'], g['B'], g['H']
    nb, p, nx, _ = A.shape
    h = hc(p, nx, B.shape[3])
    out = h.convexify_batch(A, B, H)
    for b in range(nb):
        assert int(out['status'][b]) == int(g['status'][b])
        assert rel(out['Hc'][b], g['Hc'][b]) < PARITY
        assert abs(out['kappa'][b] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])
        assert rel(out['P'][b], g['P'][b]) < PARITY          # P is as well determined as Hc (amplification <= 10: profiles/r3_secondary_outputs.txt)


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(0, 3, 3, 3, 2), (20, 4, 1, 3, 1), (30, 4, 2, 3, 1), (13, 2, 30, 4, 1), (5, 4, 16, 3, 2),
                                             (11, 2, 6, 12, 4), (12, 2, 4, 24, 8), (40, 3, 5, 2, 2), (41, 2, 7, 5, 1),
                                             (50, 1, 3, 31, 1), (51, 1, 3, 17, 3)])      # maximum n = 32 with d = 496; odd sizes (d = 153, dp = 160)
def test_parity_vs_oracle(hc, seed, nb, p, nx, mb):
    """Same seeded inputs through the HIP path and the CPU oracle (covers p=1, p=2, early-exit members)."""
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = hc(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b])
        assert int(out['status'][b]) == int(r['status'])
        assert bool(out['info'][b, 13]) == bool(r['early_exit'])
        assert rel(out['Hc'][b], r['Hc']) < PARITY
        assert abs(out['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        if not r['early_exit']:
            assert abs(out['info'][b, 0] - r['s']) < 1e-12 * r['s'] and abs(out['info'][b, 1] - r['sbeta']) < 1e-12 * r['sbeta']


def test_empty_batch_returns_empty_outputs(hc):
    h = hc(3, 3, 2, ng=1, nc=2)
    z = lambda *sh: np.zeros(sh)
    for out in (h.convexify_batch(z(0, 3, 3, 3), z(0, 3, 3, 2), z(0, 3, 5, 5)),
                h.convexify_eq_batch(z(0, 3, 3, 3), z(0, 3, 3, 2), z(0, 3, 5, 5), z(0, 3, 1, 5)),
                h.convexify_step2_batch(z(0, 3, 3, 3), z(0, 3, 3, 2), z(0, 3, 5, 5), z(0, 3, 3, 5), np.zeros((0, 3), np.int32), 1e-3)):
        assert out['Hc'].shape == (0, 3, 5, 5) and out['status'].shape == (0,) and out['kappa'].shape == (0,)


def test_scalar_fma_and_mfma_paths_agree(hc):
    A, B, H = co.gen_batch(11, 2, 6, 12, 4)
    h = hc(6, 12, 4)
    o1 = h.convexify_batch(A, B, H)
    h.set_options(flags=1)
    o2 = h.convexify_batch(A, B, H)
    h.set_options(flags=0)
    assert rel(o1['Hc'], o2['Hc']) < PARITY


def test_device_resident_entry_matches_host_entry(hc):
    import torch
    A, B, H = co.gen_batch(5, 3, 8, 3, 2)
    h = hc(8, 3, 2)
    o1 = h.convexify_batch(A, B, H)
    dev = torch.device('cuda', 0)
    o2 = h.convexify_batch_device(*(torch.from_numpy(x).to(dev) for x in (A, B, H)))
    torch.cuda.synchronize()
    assert np.array_equal(o1['Hc'], o2['Hc'].cpu().numpy())
    assert np.array_equal(o1['status'], o2['status'].cpu().numpy())


def test_chunked_batches_equal_unchunked(hc):
    """A batch larger than the workspace chunk is processed in waves; results must not depend on the chunking."""
    A, B, H = co.gen_batch(60, 7, 5, 3, 1)
    big = hc(5, 3, 1)
    small = hc(5, 3, 1, chunk=3)
    o1 = big.convexify_batch(A, B, H); o2 = small.convexify_batch(A, B, H)
    assert np.array_equal(o1['Hc'], o2['Hc']) and np.array_equal(o1['iters'], o2['iters'])


def test_infeasible_member_does_not_abort_batch(hc):
    """B = 0 with R not PD can never be convexified: status 2 for that member only."""
    A, B, H = co.gen_batch(70, 3, 2, 2, 1)
    A[1] = 0.5 * np.eye(2); B[1] = 0.0
    H[1] = co.build_hessian(np.eye(2), np.array([[-1.0]]), np.zeros((2, 1)))
    h = hc(2, 2, 1)
    out = h.convexify_batch(A, B, H)
    assert int(out['status'][1]) == 2
    for b in (0, 2):
        r = co.convexify_arrays(A[b], B[b], H[b])
        assert int(out['status'][b]) == int(r['status']) and rel(out['Hc'][b], r['Hc']) < PARITY


# ----------------------------------------------------------------------------- Step 1 with the equality-constraint term
def _eq_inputs(seed, nb, p, nx, mb, ng):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    G = np.random.default_rng(1000 + seed).standard_normal((nb, p, ng, nx + mb))
    return A, B, H, G


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng', [(20, 2, 3, 3, 2, 2), (0, 3, 3, 3, 2, 1), (30, 2, 2, 3, 1, 2), (20, 2, 1, 3, 1, 1),
                                                (7, 2, 5, 4, 2, 3), (11, 2, 4, 6, 3, 4), (5, 2, 8, 8, 4, 8), (12, 1, 3, 24, 8, 5)])
def test_equality_term_parity_vs_oracle(hc, seed, nb, p, nx, mb, ng):
    """convexifier.py:249-255 / :346-347: the multipliers Fg_k >= 0 of G_k join Step 1.  HIP path (stage-local elimination,
    tmpc_phi.h) vs the CPU oracle (border columns); p = 1, p = 2, early-exit members and the maximum ng = 8 included."""
    A, B, H, G = _eq_inputs(seed, nb, p, nx, mb, ng)
    out = hc(p, nx, mb, ng=ng).convexify_eq_batch(A, B, H, G)
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b], G=G[b])
        assert int(out['status'][b]) == int(r['status']) and bool(out['info'][b, 13]) == bool(r['early_exit'])
        assert rel(out['Hc'][b], r['Hc']) < PARITY
        assert abs(out['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        if r['early_exit']:
            assert not out['Fg'][b].any() and not out['dHc'][b].any()
            continue
        assert (out['Fg'][b] >= 0).all() and rel(out['Fg'][b], r['Fg']) < PARITY
        # the supplement is what convexHessianSuppl builds from (P, Fg): convexifier.py:196-197
        assert rel(out['dHc'][b], co.convex_hessian_suppl(A[b], B[b], out['P'][b], G=G[b], Fg=out['Fg'][b])[0]) < 1e-12
        ev = np.linalg.eigvalsh(out['Hc'][b])
        assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-8)


@pytest.mark.parametrize('name', ['eq_term_n5', 'eq_term_p1', 'eq_term_n9'])
def test_equality_term_golden_vectors(hc, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    A, B, H, G = g['A'], g['B'], g['H'], g['G']
    nb, p, nx, _ = A.shape
    out = hc(p, nx, B.shape[3], ng=G.shape[2]).convexify_eq_batch(A, B, H, G)
    for b in range(nb):
        assert int(out['status'][b]) == int(g['status'][b])
        assert rel(out['Hc'][b], g['Hc'][b]) < PARITY
        assert abs(out['kappa'][b] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])
        assert np.linalg.norm(out['Fg'][b] - g['Fg'][b]) <= PARITY * max(1.0, np.linalg.norm(g['Fg'][b]))


def test_equality_term_handle_serves_plain_calls_and_chunks(hc):
    """A handle created with ng > 0 gives the plain Step 1 bit for bit; chunked equality-term batches equal unchunked ones."""
    A, B, H, G = _eq_inputs(7, 5, 5, 4, 2, 3)
    hq = hc(5, 4, 2, ng=3)
    plain = hc(5, 4, 2).convexify_batch(A, B, H)
    np.testing.assert_array_equal(hq.convexify_batch(A, B, H)['Hc'], plain['Hc'])
    full = hq.convexify_eq_batch(A, B, H, G)
    part = hc(5, 4, 2, ng=3, chunk=2).convexify_eq_batch(A, B, H, G)
    np.testing.assert_array_equal(full['Hc'], part['Hc'])
    np.testing.assert_array_equal(full['Fg'], part['Fg'])
    assert (full['kappa'] <= plain['kappa'] * (1 + 1e-9)).all()       # the extra freedom can only lower kappa*


def test_equality_term_dense_model_vector(hc, golden_dir):
    """B = 0, R < 0: Step 1 is infeasible without G and feasible with it (the vector of the dense model, oracle/reference_sdp.py)."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(golden_dir, 'n1_step1_equality_term.npz'))
    p = g['A'].shape[0]
    lst = lambda a: [a[k] for k in range(p)]
    with pytest.raises(ValueError, match='Convexification is not possible'):
        convexifier.convexify(lst(g['A']), lst(g['B']), lst(g['Q']), lst(g['R']), lst(g['N']))
    dHc, dQc, dRc, dNc = convexifier.convexify(lst(g['A']), lst(g['B']), lst(g['Q']), lst(g['R']), lst(g['N']), G=lst(g['Cu']))
    H = np.stack([co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) for k in range(p)])
    ev = np.linalg.eigvalsh(H + np.stack(dHc))
    assert ev.min() > 0 and abs((ev[:, -1] / ev[:, 0]).max() / float(g['kappa']) - 1.0) < 1e-4
    assert all(dRc[k][0, 0] > 0.5 for k in range(p))


# ----------------------------------------------------------------------------- Step 2: active-constraint multipliers + norm terms
def _step2_inputs(seed, nb, p, nx, mb, ng, ncs):
    n = nx + mb
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 5)
    nc = max(max(ncs), 1)
    G = rng.standard_normal((nb, p, ng, n))
    C = np.zeros((nb, p, nc, n)); ncnt = np.tile(np.asarray(ncs, np.int32), (nb, 1))
    for b in range(nb):
        for k in range(p):
            C[b, k, :ncs[k]] = rng.standard_normal((ncs[k], n))
    return A, B, H, G, C, ncnt


def _step2_oracle(A, B, H, G, C, ncnt, rho):
    p = A.shape[0]
    Cl = [C[k, :ncnt[k]] if ncnt[k] else None for k in range(p)]
    Gb = G if G.shape[1] else None
    r = co.sdp_step1(A, B, H, G=Gb, C=Cl, rho=rho)
    st, dHc = co.check_convergence(A, B, H, r['P'], r['ipm_status'], G=Gb, Fg=r.get('Fg'), C=Cl, F=r['F'])[:2]
    return r, st, dHc, Cl


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng,ncs,rho', [
    (20, 2, 3, 3, 2, 0, [2, 0, 1], 1e-3), (0, 1, 3, 3, 2, 2, [1, 2, 0], 1.0), (30, 2, 2, 3, 1, 1, [1, 1], 1e-3), (20, 1, 1, 3, 1, 0, [2], 1.0),
    (7, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3], 1e-2), (11, 2, 4, 6, 3, 2, [4, 0, 8, 1], 1e-3), (12, 1, 3, 24, 8, 4, [8, 3, 0], 1e-3),
    (52, 1, 3, 31, 1, 8, [8, 5, 0], 1e-3),           # the largest block: d = 496 plus 18 multipliers (dp = 528)
    (14, 1, 3, 6, 2, 12, [16, 9, 0], 1e-2),          # the most multipliers: 12 + 16 rows and two norm terms of 12 and 16
    (13, 1, 3, 8, 2, 2, [0, 0, 0], 1e-2)])
def test_step2_parity_vs_oracle(hc, seed, nb, p, nx, mb, ng, ncs, rho):
    """convexifier.py:116-131 (constr=True): multipliers F_k >= 0 of ragged active-constraint Jacobians (stages without C_k
    included), the norm terms rho*||F_k|| and rho*||Fg_k|| as arrow LMIs.  HIP path (stage-local elimination, tmpc_phi.h) vs the
    CPU oracle (border columns); p = 1, p = 2, the maximum row counts and a case with G only (every C_k None)."""
    A, B, H, G, C, ncnt = _step2_inputs(seed, nb, p, nx, mb, ng, ncs)
    J = np.concatenate([G, C], axis=2)
    out = hc(p, nx, mb, ng=ng, nc=C.shape[2]).convexify_step2_batch(A, B, H, J, ncnt, rho)
    for b in range(nb):
        r, st, dHc, Cl = _step2_oracle(A[b], B[b], H[b], G[b], C[b], ncnt[b], rho)
        assert int(out['status'][b]) == int(st)
        assert rel(out['Hc'][b], H[b] + dHc) < PARITY
        assert abs(out['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        Fo = out['FgF'][b]
        assert (Fo >= 0).all()
        for k in range(p):
            if ng:
                assert np.linalg.norm(Fo[k, :ng] - r['Fg'][k]) <= PARITY * max(1.0, np.linalg.norm(r['Fg'][k]))
            if ncnt[b, k]:
                assert np.linalg.norm(Fo[k, ng:ng + ncnt[b, k]] - r['F'][k]) <= PARITY * max(1.0, np.linalg.norm(r['F'][k]))
            assert not Fo[k, ng + ncnt[b, k]:].any()
        # the supplement is what convexHessianSuppl builds from (P, Fg, F): convexifier.py:196-201
        Fl = [Fo[k, ng:ng + ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        ref = co.convex_hessian_suppl(A[b], B[b], out['P'][b], G=G[b] if ng else None, Fg=Fo[:, :ng] if ng else None, C=Cl, F=Fl)[0]
        assert rel(out['dHc'][b], ref) < 1e-12


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng,ncs', [(20, 2, 3, 3, 2, 0, [2, 0, 1]), (7, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3]), (11, 1, 4, 6, 3, 2, [4, 0, 8, 1]),
                                                    (30, 1, 2, 3, 1, 1, [1, 1]), (12, 1, 3, 24, 8, 4, [8, 3, 0])])
def test_step2_beta_only_objective_parity(hc, seed, nb, p, nx, mb, ng, ncs):
    """The OTHER reading of convexifier.py:276-283 (`picos.sum(obj, abs(rho*F[i]))` may drop its second argument in PICOS 1.2.0, SURVEY.md
    7.0): objective beta alone, multipliers F_k, Fg_k cost-free.  rho = 0 through the C ABI vs the oracle's cost_free model; ragged C_k."""
    A, B, H, G, C, ncnt = _step2_inputs(seed, nb, p, nx, mb, ng, ncs)
    J = np.concatenate([G, C], axis=2)
    out = hc(p, nx, mb, ng=ng, nc=C.shape[2]).convexify_step2_batch(A, B, H, J, ncnt, 0.0)
    for b in range(nb):
        Cl = [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        Gb = G[b] if ng else None
        r = co.sdp_step1(A[b], B[b], H[b], G=Gb, C=Cl, cost_free=True)
        st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], G=Gb, Fg=r.get('Fg'), C=Cl, F=r['F'])[:2]
        assert int(out['status'][b]) == int(st)
        assert rel(out['Hc'][b], H[b] + dHc) < PARITY
        assert abs(out['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        Fo = out['FgF'][b]
        assert (Fo >= 0).all() and all(not Fo[k, ng + ncnt[b, k]:].any() for k in range(p))
        # cost-free multipliers can only lower kappa compared with the paper's objective at any rho > 0
    paper = hc(p, nx, mb, ng=ng, nc=C.shape[2]).convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
    assert (out['kappa'] <= paper['kappa'] * (1 + 1e-7)).all()


@pytest.mark.parametrize('name', ['step2_ragged_n5', 'step2_with_g_n6', 'step2_p1'])
def test_step2_golden_vectors(hc, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    A, B, H, G, C, ncnt = g['A'], g['B'], g['H'], g['G'], g['C'], g['ncnt']
    nb, p, nx, _ = A.shape
    ng = G.shape[2]
    out = hc(p, nx, B.shape[3], ng=ng, nc=C.shape[2]).convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, float(g['rho']))
    for b in range(nb):
        assert int(out['status'][b]) == int(g['status'][b])
        assert rel(out['Hc'][b], g['Hc'][b]) < PARITY
        assert abs(out['kappa'][b] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])
        assert np.linalg.norm(out['FgF'][b][:, ng:] - g['F'][b]) <= PARITY * max(1.0, np.linalg.norm(g['F'][b]))
        if ng:
            assert np.linalg.norm(out['FgF'][b][:, :ng] - g['Fg'][b]) <= PARITY * max(1.0, np.linalg.norm(g['Fg'][b]))


def test_device_resident_constraint_entries_match_host_entries(hc):
    """tmpc_convexify_con_batch_device (torch tensors in HBM) == the host-buffer entries, for Step 1 with G and for Step 2."""
    import torch
    A, B, H, G, C, ncnt = _step2_inputs(7, 3, 5, 4, 2, 3, [0, 3, 1, 2, 3])
    J = np.concatenate([G, C], axis=2)
    h = hc(5, 4, 2, ng=3, nc=C.shape[2])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ref = h.convexify_eq_batch(A, B, H, G)
    dev = h.convexify_con_batch_device(t(A), t(B), t(H), t(G))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dev['Hc'].cpu().numpy(), ref['Hc']); np.testing.assert_array_equal(dev['FgF'].cpu().numpy(), ref['Fg'])
    ref = h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
    dev = h.convexify_con_batch_device(t(A), t(B), t(H), t(J), ncnt=t(ncnt), rho=1e-2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dev['Hc'].cpu().numpy(), ref['Hc']); np.testing.assert_array_equal(dev['FgF'].cpu().numpy(), ref['FgF'])
    np.testing.assert_array_equal(dev['status'].cpu().numpy(), ref['status'])
    # chunked Step 2 batches (3 problems through a 2-problem workspace) equal unchunked ones
    part = hc(5, 4, 2, ng=3, nc=C.shape[2], chunk=2).convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
    np.testing.assert_array_equal(part['Hc'], ref['Hc']); np.testing.assert_array_equal(part['FgF'], ref['FgF'])


def test_step_logic_for_a_batch(golden_dir):
    """convexify_steps_batch: Step 1 for all, Step 2 only for the members Step 1 cannot convexify (here the dense model's vector,
    B = 0 and R < 0), next to a member that Step 1 solves and one that is already convex."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(golden_dir, 'n1_step2_active_constraints.npz'))
    p, nx = g['A'].shape[0], g['A'].shape[1]
    Hr = np.stack([co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) for k in range(p)])
    A1, B1, H1 = co.gen_batch(0, 2, p, nx, g['B'].shape[2])                # member 0 (seed 0): indefinite, Step 1 suffices
    A = np.stack([A1[0], g['A'], A1[1]]); B = np.stack([B1[0], g['B'], B1[1]]); H = np.stack([H1[0], Hr, np.stack([np.eye(nx + 1)] * p)])
    C = np.stack([g['Cu']] * 3); ncnt = np.full((3, p), g['Cu'].shape[1], np.int32)
    out = convexifier.convexify_steps_batch(A, B, H, C=C, ncnt=ncnt, rho=float(g['rho']))
    assert list(out['step']) == [1, 2, 0] and list(out['status']) == [0, 0, 0]
    r0 = co.convexify_arrays(A[0], B[0], H[0])
    r1 = co.convexify_arrays(A[1], B[1], H[1], C=[c for c in g['Cu']], rho=float(g['rho']))
    assert rel(out['Hc'][0], r0['Hc']) < PARITY and rel(out['Hc'][1], r1['Hc']) < PARITY and r1['step'] == 2
    assert not out['F'][0].any() and not out['F'][2].any() and (out['F'][1] > 0).all() and not out['dHc'][2].any()


def test_step2_dropin_takes_over_when_step1_is_infeasible(golden_dir):
    """convexify(..., C=...) on the dense model's vector (B = 0, R < 0): Step 1 infeasible -> Step 2 -> EQUIVALENCE TYPE B;
    without C the reference's ValueError."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(golden_dir, 'n1_step2_active_constraints.npz'))
    p = g['A'].shape[0]
    lst = lambda a: [a[k] for k in range(p)]
    args = (lst(g['A']), lst(g['B']), lst(g['Q']), lst(g['R']), lst(g['N']))
    with pytest.raises(ValueError, match='Convexification is not possible'):
        convexifier.convexify(*args)
    forced = convexifier.convexify(*args, opts={'rho': 1e-3, 'force': True})          # no constraints given: Step 3 on the plain model
    assert all(np.linalg.eigvalsh(co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) + forced[0][k]).min() > 0 for k in range(p))
    dHc, dQc, dRc, dNc = convexifier.convexify(*args, C=lst(g['Cu']), opts={'rho': float(g['rho'])})
    ref = co.convexify(*args, C=lst(g['Cu']), opts={'rho': float(g['rho'])})
    H = np.stack([co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) for k in range(p)])
    assert rel(H + np.stack(dHc), H + np.stack(ref[0])) < PARITY
    ev = np.linalg.eigvalsh(H + np.stack(dHc))
    assert ev.min() > 0 and abs((ev[:, -1] / ev[:, 0]).max() / float(g['kappa']) - 1.0) < 1e-4
    assert all(dRc[k][0, 0] > 0.5 for k in range(p))
    # a stage without active constraints (None entry, preprocessing.py:180) and a ragged list
    Cr = [g['Cu'][0], None] if p == 2 else lst(g['Cu'])
    out = convexifier.convexify(*args, C=[np.vstack([g['Cu'][0], [[1.0, 0.0, 0.0]]]), g['Cu'][1]], opts={'rho': float(g['rho'])})
    assert np.linalg.eigvalsh(H + np.stack(out[0])).min() > 0
    with pytest.raises(ValueError, match='Convexification is not possible'):
        convexifier.convexify(*args, C=Cr, opts={'rho': float(g['rho'])})        # stage 1 keeps R < 0 without its constraint


@pytest.mark.parametrize('seed,p,nx,mb,gam', [(61, 3, 3, 2, 50.0), (62, 4, 4, 2, 50.0), (63, 2, 3, 1, 20.0)])
def test_step2_dropin_beta_only_objective(seed, p, nx, mb, gam):
    """opts={'objective': 'beta'}: the other reading of convexifier.py:276-285 through the drop-in.  The stage Hessians of a feasible problem
    are made indefinite along ragged constraint rows, H_k -= gamma C_k' C_k, so that Step 1 is infeasible and Step 2 has to find
    F_k ~ gamma: the drop-in with cost-free multipliers against the oracle's cost_free model; the paper's objective can only give a larger
    condition number; with 'force' the drop-in refuses (a cost-free T_k has no counterpart); unknown values raise."""
    from tunempc_amd import convexifier
    A, B, H = co.gen_batch(seed, 1, p, nx, mb)
    A, B, H = A[0], B[0], H[0].copy()
    rng = np.random.default_rng(seed + 1)
    C = [rng.standard_normal((1 if k % 2 else 2, nx + mb)) for k in range(p)]
    for k in range(p):
        H[k] -= gam * C[k].T @ C[k]
    lst = lambda a: [a[k] for k in range(p)]
    args = (lst(A), lst(B), [H[k][:nx, :nx] for k in range(p)], [H[k][nx:, nx:] for k in range(p)], [H[k][:nx, nx:] for k in range(p)])
    with pytest.raises(ValueError, match='Convexification is not possible'):
        convexifier.convexify(*args)                                           # Step 1 alone cannot do it
    dHc = convexifier.convexify(*args, C=C, opts={'objective': 'beta'})[0]
    r = co.sdp_step1(A, B, H, C=C, cost_free=True)
    ref = co.convex_hessian_suppl(A, B, r['P'], C=C, F=r['F'])[0]
    assert r['ipm_status'] == 'optimal' and rel(H + np.stack(dHc), H + ref) < PARITY
    paper = convexifier.convexify(*args, C=C, opts={'rho': 1e-3})[0]
    cond = lambda d: (lambda ev: (ev[:, -1] / ev[:, 0]).max())(np.linalg.eigvalsh(H + np.stack(d)))
    assert cond(dHc) <= cond(paper) * (1 + 1e-6)
    with pytest.raises(ValueError, match='unknown objective'):
        convexifier.convexify(*args, C=C, opts={'objective': 'gamma'})
    with pytest.raises(NotImplementedError, match='beta'):
        convexifier.convexify(*args, C=[None] * p, opts={'objective': 'beta', 'force': True})    # no active rows at all: still infeasible after Step 2


# ----------------------------------------------------------------------------- reference-compatible API
def test_dropin_convexify_lqr_example():
    """examples/convex_lqr.py through the drop-in API: same call, same return structure, same assertion (:58)."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'c1_convex_lqr.npz'))
    A = np.matrix(g['A'][0, 0]); B = np.matrix(g['B'][0, 0]); Q = np.matrix(g['Q']); R = np.matrix(g['R']); N = np.matrix(g['N'])
    dHc, dQc, dRc, dNc = convexifier.convexify(A, B, Q, R, N)
    assert isinstance(dHc, list) and len(dHc) == 1 and dHc[0].shape == (4, 4)
    An, Bn, Qn, Rn, Nn = (np.asarray(x) for x in (A, B, Q, R, N))

    def gain(Q_, R_, N_):
        P = sla.solve_discrete_are(An, Bn, Q_, R_, s=N_)
        return np.linalg.solve(R_ + Bn.T @ P @ Bn, Bn.T @ P @ An + N_.T)
    assert np.linalg.norm(gain(Qn, Rn, Nn) - gain(Qn + dQc[0], Rn + dRc[0], Nn + dNc[0])) < 1e-5
    assert rel(np.asarray(dHc[0]) + co.build_hessian(Qn, Rn, Nn), g['Hc'][0, 0]) < PARITY


def test_dropin_solver_log_at_debug_level(caplog):
    """convexifier.py:87-91: below INFO level the solver is verbose -- here the iteration log of the device IPM goes to the logger."""
    import logging
    from tunempc_amd import convexifier
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'c1_convex_lqr.npz'))
    args = (g['A'][0, 0], g['B'][0, 0], g['Q'], g['R'], g['N'])
    with caplog.at_level(logging.INFO, logger='tunempc'):
        convexifier.convexify(*args)
    assert not any('rel.step' in r.message for r in caplog.records) and any('Optimal solution found.' in r.message for r in caplog.records)
    caplog.clear()
    with caplog.at_level(logging.DEBUG, logger='tunempc'):
        convexifier.convexify(*args)
    msgs = [r.message for r in caplog.records]
    its = [m for m in msgs if m.split() and m.split()[0].isdigit()]
    assert any('rel.step' in m for m in msgs) and len(its) >= 5 and any('center' in m for m in its)


def test_dropin_early_exit_and_infeasible():
    from tunempc_amd import convexifier
    rng = np.random.default_rng(3)
    A = rng.standard_normal((3, 3)); B = rng.standard_normal((3, 1))
    out = convexifier.convexify(A, B, 2 * np.eye(3), 2 * np.eye(1), np.zeros((3, 1)))
    assert all(isinstance(o, np.ndarray) for o in out) and out[0].shape == (4, 4) and not out[0].any()   # convexifier.py:85
    with pytest.raises(ValueError, match='Convexification is not possible'):
        convexifier.convexify(0.5 * np.eye(2), np.zeros((2, 1)), np.eye(2), np.array([[-1.0]]), np.zeros((2, 1)))


def test_dropin_tuner_convexify_list_inputs():
    """Tuner.convexify slicing (tuner.py:145-158) over a sensitivities dict."""
    from tunempc_amd.tuner import tuner_convexify
    A, B, H = co.gen_batch(2000, 1, 30, 4, 1)
    S = {'A': [A[0, k] for k in range(30)], 'B': [B[0, k] for k in range(30)], 'H': [H[0, k] for k in range(30)], 'C_As': None, 'G': None}
    Hc = tuner_convexify(S, nx=4, p=30)
    r = co.convexify_arrays(A[0], B[0], H[0])
    assert len(Hc) == 30 and rel(np.stack(Hc), r['Hc']) < PARITY


# ----------------------------------------------------------------------------- Step 3: forced regularisation T (plain model + T)
@pytest.mark.parametrize('seed,nb,p,nx,mb,rho', [(0, 2, 2, 2, 1, 1e-3), (1, 2, 3, 3, 2, 1e-2), (2, 2, 1, 3, 1, 1e-3), (6, 1, 4, 4, 3, 1.0), (4, 1, 2, 9, 6, 1e-3),
                                                 (5, 1, 5, 2, 2, 1e-1)])
def test_step3_parity_vs_oracle(hc, seed, nb, p, nx, mb, rho):
    """The Step 3 model (convexifier.py:137-147: T_k symmetric, every entry > 0, rho*||T_k||_F in the objective) through the HIP path
    and the structured oracle (same model, norm term as a second-order cone), solved directly on seeded problems."""
    A, B, H = co.gen_batch(300 + seed, nb, p, nx, mb)
    h = hc(p, nx, mb, step3=True)
    out = h.convexify_step3_batch(A, B, H, rho)
    convex = [min(np.linalg.eigvalsh(co.symmetrize(H[b, k])).min() for k in range(p)) > 0 for b in range(nb)]
    assert [bool(v != 0.0) for v in out['info'][:, 13]] == convex and not all(convex)     # members convex already are returned untouched
    for b in range(nb):
        if convex[b]:
            assert rel(out['Hc'][b], H[b]) < 1e-15
            continue
        r = co.sdp_step1(A[b], B[b], H[b], rho=rho, force=True)
        st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], T=r['T'])[:2]
        assert int(out['status'][b]) == st == 0
        assert rel(out['Hc'][b], H[b] + dHc) < PARITY
        assert abs(out['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        assert rel(out['T'][b], r['T']) < PARITY and (out['T'][b] > 0).all()
        # structure of the supplement: Hc - H = sym(calH(P)) + T exactly
        ref = co.convex_hessian_suppl(A[b], B[b], out['P'][b], T=out['T'][b])[0]
        assert rel(out['Hc'][b] - H[b], ref) < 1e-12


@pytest.mark.parametrize('seed,p,nx,mb,ng,ncs,rho', [(4, 2, 2, 1, 1, None, 1e-2), (1, 3, 3, 2, 1, [2, 0, 1], 1e-2), (3, 1, 3, 1, 0, [2], 1e-1),
                                                    (3, 4, 4, 2, 2, [1, 3, 0, 2], 1e-3), (4, 2, 6, 3, 1, [2, 2], 1e-2),
                                                    (5, 2, 8, 4, 18, [20, 17], 1e-2)])      # (more than 32 rows per stage: Gram products in global memory)
def test_step3_with_constraint_rows_parity(hc, seed, p, nx, mb, ng, ncs, rho):
    """Step 3 with the multipliers of G / C in the same solve (convexifier.py:144: constr = constraint_contribution, force = True) against
    the structured oracle: Hc, the multipliers Fg / F, T and the objective; ncs None: G only (the reference's constr = False)."""
    n = nx + mb
    A, B, H = co.gen_batch(700 + seed, 1, p, nx, mb)
    rng = np.random.default_rng(900 + seed)
    G = rng.standard_normal((1, p, ng, n)) if ng else None
    nc = max(ncs) if ncs else 0
    h = hc(p, nx, mb, ng=ng, nc=nc, step3=True)
    if ncs is None:
        out = h.convexify_step3_con_batch(A, B, H, G, None, rho)
        r = co.sdp_step1(A[0], B[0], H[0], G=G[0], rho=rho, force=True)
        Cl = None
    else:
        C = np.zeros((1, p, nc, n))
        for k in range(p):
            C[0, k, :ncs[k]] = rng.standard_normal((ncs[k], n))
        J = C if not ng else np.concatenate([G, C], axis=2)
        out = h.convexify_step3_con_batch(A, B, H, J, np.asarray([ncs], np.int32), rho)
        Cl = [C[0, k, :ncs[k]] if ncs[k] else None for k in range(p)]
        r = co.sdp_step1(A[0], B[0], H[0], G=None if G is None else G[0], C=Cl, rho=rho, force=True)
    assert out['info'][0, 13] == 0.0                     # (not already convex: an SDP was solved)
    assert int(out['status'][0]) == 0 and r['ipm_status'] == 'optimal'
    dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], G=None if G is None else G[0], Fg=r.get('Fg'), C=Cl, F=r.get('F'), T=r['T'])[0]
    assert rel(out['Hc'][0], H[0] + dHc) < PARITY
    assert abs(out['kappa'][0] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
    assert rel(out['T'][0], r['T']) < PARITY and (out['T'][0] > 0).all()
    if ng:
        assert np.abs(out['FgF'][0, :, :ng] - r['Fg']).max() < PARITY * max(1.0, np.abs(r['Fg']).max())
    if ncs:
        for k in range(p):
            if ncs[k]:
                assert np.abs(out['FgF'][0, k, ng:ng + ncs[k]] - r['F'][k]).max() < PARITY * max(1.0, np.abs(r['F'][k]).max())


def test_step3_device_resident_entries(hc):
    """tmpc_convexify_step3_batch_device / tmpc_convexify_step3_con_batch_device (round 4): torch tensors in HBM in and out, bit-identical to the host-buffer
    entries on the same inputs (same kernels, same launch sequence; only the copies differ)."""
    import torch
    dev = torch.device('cuda', 0)
    p, nx, mb, nb, ng, nc, rho = 4, 4, 2, 3, 1, 2, 1e-2
    n = nx + mb
    A, B, H = co.gen_batch(7300, nb, p, nx, mb)
    rng = np.random.default_rng(73)
    G = rng.standard_normal((nb, p, ng, n)); Cc = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            Cc[b, k, ncnt[b, k]:] = 0.0
    J = np.concatenate([G, Cc], axis=2)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    h3 = hc(p, nx, mb, step3=True)
    a = h3.convexify_step3_batch(A, B, H, rho)
    d = h3.convexify_step3_batch_device(t(A), t(B), t(H), rho)
    torch.cuda.synchronize()
    for k in ('Hc', 'T', 'P', 'kappa', 'status', 'iters'):
        assert np.array_equal(a[k], d[k].cpu().numpy()), k
    h3c = hc(p, nx, mb, ng=ng, nc=nc, step3=True)
    a = h3c.convexify_step3_con_batch(A, B, H, J, ncnt, rho)
    d = h3c.convexify_step3_batch_device(t(A), t(B), t(H), rho, J=t(J), ncnt=t(ncnt))
    torch.cuda.synchronize()
    for k in ('Hc', 'T', 'P', 'FgF', 'kappa', 'status', 'iters'):
        assert np.array_equal(a[k], d[k].cpu().numpy()), k
    assert (a['status'] == 0).all()


def test_step3_rescues_the_infeasible_vector(golden_dir):
    """The dense model's vector (B = 0, R < 0, no constraints): Steps 1 and 2 cannot help, force -> Step 3 (convexifier.py:137-147).
    The drop-in logs the reference's warnings, takes Step 3 on the GPU and returns a positive definite result; its optimal value
    agrees with the dense restatement's (the minimiser is not unique in T beyond the objective)."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(golden_dir, 'n1_step3_force.npz'))
    A, B, Q, R, N = ([m for m in g[k]] for k in ('A', 'B', 'Q', 'R', 'N'))
    with pytest.raises(ValueError):
        convexifier.convexify(A, B, Q, R, N, opts={'rho': float(g['rho']), 'solver': 'hip', 'force': False})
    dHc, dQc, dRc, dNc = convexifier.convexify(A, B, Q, R, N, opts={'rho': float(g['rho']), 'solver': 'hip', 'force': True})
    H = [co.build_hessian(q, r, n_) for q, r, n_ in zip(Q, R, N)]
    for k in range(len(A)):
        assert np.linalg.eigvalsh(H[k] + dHc[k]).min() > 0
    out = convexifier.convexify_step3_batch(np.stack(A)[None], np.stack(B)[None], np.stack(H)[None], float(g['rho']))
    assert int(out['status'][0]) == 0 and abs(out['kappa'][0] - float(g['kappa'])) < 1e-5 * float(g['kappa'])


def test_step3_dropin_with_constraint_rows(golden_dir):
    """The same vector through convexify(..., G=, C=, opts={'force': True}): Steps 1 and 2 stay infeasible (the Jacobian rows do not reach
    the negative R block), Step 3 takes the multipliers along (convexifier.py:144) and returns a positive definite result."""
    from tunempc_amd import convexifier
    g = np.load(os.path.join(golden_dir, 'n1_step3_force.npz'))
    A, B, Q, R, N = ([m for m in g[k]] for k in ('A', 'B', 'Q', 'R', 'N'))
    p, nx, nu = len(A), A[0].shape[0], B[0].shape[1]
    rng = np.random.default_rng(5)
    G = [np.hstack([rng.standard_normal((1, nx)), np.zeros((1, nu))]) for _ in range(p)]          # rows without an input part: no help for R < 0
    C = [np.hstack([rng.standard_normal((2, nx)), np.zeros((2, nu))]) if k % 2 == 0 else None for k in range(p)]
    with pytest.raises(ValueError):
        convexifier.convexify(A, B, Q, R, N, G=G, C=C, opts={'rho': float(g['rho']), 'solver': 'hip', 'force': False})
    dHc, dQc, dRc, dNc = convexifier.convexify(A, B, Q, R, N, G=G, C=C, opts={'rho': float(g['rho']), 'solver': 'hip', 'force': True})
    for k in range(p):
        assert np.linalg.eigvalsh(co.build_hessian(Q[k], R[k], N[k]) + dHc[k]).min() > 0


def test_step3_handle_serves_the_plain_model(hc):
    A, B, H = co.gen_batch(5, 2, 4, 3, 2)
    o1 = hc(4, 3, 2).convexify_batch(A, B, H)
    o2 = hc(4, 3, 2, step3=True).convexify_batch(A, B, H)
    assert np.array_equal(o1['Hc'], o2['Hc'])


# ----------------------------------------------------------------------------- full-size parity against the compiled CPU port of the oracle
@pytest.mark.parametrize('name,p,nx,mb,nb', [('c4 bench shape', 64, 24, 8, 2), ('c5 AWE-shaped synthetic', 200, 20, 10, 1)])
def test_full_size_parity_vs_cpu_port(hc, name, p, nx, mb, nb):
    """BASELINE configs[3] / configs[4] at their FULL stage size and period, against oracle/cpu_ipm -- the C++/OpenMP restatement of the
    structured oracle (tests/test_cpu_ipm.py ties it to the numpy oracle at <= 1.6e-9) -- which solves such a problem in seconds where
    the numpy oracle needs minutes: the same 1e-8 bar as every other parity test, on the headline workload itself."""
    import cpu_ipm
    from tunempc_amd import synthetic
    A, B, H = synthetic.gen_batch(424242, nb, p, nx, mb)
    out = hc(p, nx, mb).convexify_batch(A, B, H)
    ref = cpu_ipm.convexify_batch(A, B, H, threads=min(nb, 4))
    for b in range(nb):
        assert int(out['status'][b]) == int(ref['status'][b]) == 0
        assert rel(out['Hc'][b], ref['Hc'][b]) < PARITY
        assert abs(out['kappa'][b] - ref['kappa'][b]) < 1e-9 * max(1.0, ref['kappa'][b])


# ----------------------------------------------------------------------------- full-size properties (no oracle)
def test_full_size_c4_properties(hc):
    """BASELINE configs[3] stage size and period (nx=24, m=8, p=64), a small batch: size-independent properties --
    Hc positive definite, cond(Hc_k) <= kappa, Hc - H has exactly the structure of eq. (18) for the returned P,
    kappa no worse than the feasible point the generator hides, all members converge."""
    from tunempc_amd import synthetic
    p, nx, mb, nb = 64, 24, 8, 3
    A, B, H = synthetic.gen_batch(777, nb, p, nx, mb)
    h = hc(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    assert (out['status'] == 0).all()
    for b in range(nb):
        ev = np.linalg.eigvalsh(out['Hc'][b])
        assert ev.min() > 0
        assert (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-9)
        dH, _, _, _ = co.convex_hessian_suppl(A[b], B[b], out['P'][b])
        assert rel(out['Hc'][b] - H[b], dH) < 1e-10
        assert out['kappa'][b] <= 10.0 * (1 + 1e-6)          # the generator's Hhat has cond <= 10
    # idempotence of the pre-check: convexified Hessians are already convex -> early exit, zero supplement
    again = h.convexify_batch(A, B, out['Hc'])
    assert (again['info'][:, 13] == 1).all() and not again['dHc'].any()


def test_full_size_step2_properties(hc):
    """The Step 2 model and Step 1 with G at the benchmark stage size and period (nx=24, m=8, p=64; the oracle needs minutes
    there): size-independent properties -- every member converges, Hc positive definite, cond(Hc_k) <= kappa, the supplement is
    exactly calH(P) + G' diag(Fg) G + C' diag(F) C for the returned (P, Fg, F), multipliers >= 0 and zero in the padding, the extra
    freedom can only lower kappa, and a larger rho can only raise it.  (At p = 64 the first start point of the norm terms made
    this model diverge while every small parity case passed.)"""
    from tunempc_amd import synthetic
    p, nx, mb, nb, ng, nc = 64, 24, 8, 4, 2, 4
    n = nx + mb
    A, B, H = synthetic.gen_batch(777, nb, p, nx, mb)
    rng = np.random.default_rng(3)
    G = rng.standard_normal((nb, p, ng, n)); C = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            C[b, k, ncnt[b, k]:] = 0.0
    h = hc(p, nx, mb, ng=ng, nc=nc)
    plain = h.convexify_batch(A, B, H)
    eq = h.convexify_eq_batch(A, B, H, G)
    kap = {}
    for rho in (1e-3, 1.0):
        out = h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, rho)
        assert (out['status'] == 0).all() and out['iters'].max() <= 30
        kap[rho] = out['kappa']
        for b in range(nb):
            ev = np.linalg.eigvalsh(out['Hc'][b])
            assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-8)
            Fo = out['FgF'][b]
            assert (Fo >= 0).all() and all(not Fo[k, ng + ncnt[b, k]:].any() for k in range(p))
            Cl = [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
            Fl = [Fo[k, ng:ng + ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
            ref = co.convex_hessian_suppl(A[b], B[b], out['P'][b], G=G[b], Fg=Fo[:, :ng], C=Cl, F=Fl)[0]
            assert rel(out['Hc'][b] - H[b], ref) < 1e-10
    assert (eq['status'] == 0).all() and (eq['kappa'] <= plain['kappa'] * (1 + 1e-9)).all()
    assert (kap[1e-3] <= kap[1.0] * (1 + 1e-7)).all()
    for b in range(nb):
        ref = co.convex_hessian_suppl(A[b], B[b], eq['P'][b], G=G[b], Fg=eq['Fg'][b])[0]
        assert rel(eq['Hc'][b] - H[b], ref) < 1e-10 and (eq['Fg'][b] >= 0).all()


def test_c5_golden_p200_n30(hc, golden_dir):
    """BASELINE configs[4] (AWE-shaped synthetic: p=200, nx=20, m=10, n=30, d=210): HIP path vs the committed oracle vector."""
    g = np.load(os.path.join(golden_dir, 'c5_awe_synthetic_p200_n30.npz'))
    A, B, H = g['A'], g['B'], g['H']
    nb, p, nx, _ = A.shape
    assert (p, nx, B.shape[3]) == (200, 20, 10)
    h = hc(p, nx, B.shape[3], chunk=2)
    out = h.convexify_batch(A, B, H)
    for b in range(nb):
        assert int(out['status'][b]) == int(g['status'][b]) == 0
        assert rel(out['Hc'][b], g['Hc'][b]) < PARITY
        assert abs(out['kappa'][b] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])


def test_full_size_c5_properties(hc):
    """BASELINE configs[4] full shape (p=200, nx=20, m=10), a small batch (the published 512 shard to 64 per GPU; the
    properties do not depend on the batch size): every member converges, Hc positive definite, cond(Hc_k) <= kappa,
    Hc - H = sym(calH(P)) for the returned P, kappa no worse than the generator's hidden feasible point, idempotence."""
    from tunempc_amd import synthetic
    p, nx, mb, nb = 200, 20, 10, 3
    A, B, H = synthetic.gen_batch(555, nb, p, nx, mb)
    h = hc(p, nx, mb, chunk=4)
    out = h.convexify_batch(A, B, H)
    assert (out['status'] == 0).all() and out['iters'].max() <= 30
    for b in range(nb):
        ev = np.linalg.eigvalsh(out['Hc'][b])
        assert ev.min() > 0
        assert (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-9)
        dH, _, _, _ = co.convex_hessian_suppl(A[b], B[b], out['P'][b])
        assert rel(out['Hc'][b] - H[b], dH) < 1e-10
        assert out['kappa'][b] <= 10.0 * (1 + 1e-6)
    again = h.convexify_batch(A, B, out['Hc'])
    assert (again['info'][:, 13] == 1).all() and not again['dHc'].any()


def test_sharded_path_with_hip_handle_and_nccl(hc):
    """tunempc_amd.dist.convexify_batch_sharded with the REAL HIP handle as solve_fn and the nccl (= RCCL) backend, world
    size 1 (the GPU box has one device): the code path bench.py --gpus N runs per rank, checked against the plain call."""
    import torch
    import torch.distributed as dist
    from tunempc_amd.dist import convexify_batch_sharded
    A, B, H = co.gen_batch(910, 5, 4, 3, 2)
    h = hc(4, 3, 2)
    ref = h.convexify_batch(A, B, H)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 300))
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        dA, dB, dH = (torch.from_numpy(x).to(dev) for x in (A, B, H))
        g = convexify_batch_sharded(dA, dB, dH, lambda a, b, c: h.convexify_batch_device(a.contiguous(), b.contiguous(), c.contiguous()))
        torch.cuda.synchronize()
        assert np.array_equal(g['Hc'].cpu().numpy(), ref['Hc'])
        assert np.array_equal(g['status'].cpu().numpy(), ref['status'])
        assert np.array_equal(g['kappa'].cpu().numpy(), ref['kappa'])
    finally:
        if own:
            dist.destroy_process_group()


def _certificate(A, B, H, out, dual, b):
    """Solver-independent optimality certificate of member b, in numpy only (no oracle): the primal point returned by the library
    (P, alpha, kappa) is feasible for the scaled SDP of convexifier.py:304-306 with objective kappa; the exported dual iterate is
    feasible up to tiny residuals and its objective is a lower bound of the optimal kappa* (weak duality).  Returns
    (kappa, dual objective, slack from the dual residuals, N * mu_target)."""
    p, nx, _ = A.shape
    n = H.shape[1]
    s = out['info'][b, 0]
    alpha = out['alpha'][b]; kappa = out['kappa'][b]
    Hb = s * 0.5 * (H + H.transpose(0, 2, 1))
    V = np.concatenate([A, B], axis=2)
    Pbar = out['P'][b] * (s * alpha)                                        # un-scaling of convexifier.py:406 undone
    M = alpha * Hb + V.transpose(0, 2, 1) @ np.roll(Pbar, -1, axis=0) @ V
    M[:, :nx, :nx] -= Pbar
    ev = np.linalg.eigvalsh(M)
    assert ev.min() >= 1.0 - 1e-9 and ev.max() <= kappa * (1 + 1e-9) and alpha > 1e-8       # primal feasible, objective kappa
    X1, X2, x0 = dual['X1'][b], dual['X2'][b], dual['x0'][b]
    assert np.linalg.eigvalsh(X1).min() > 0 and np.linalg.eigvalsh(X2).min() > 0 and x0 > 0
    Y = X1 - X2
    r_tau = 1.0 - np.trace(X2, axis1=1, axis2=2).sum()
    r_alpha = -np.sum(Hb * Y) - x0
    W = V @ Y @ V.transpose(0, 2, 1)
    r_P = -(np.roll(W, 1, axis=0) - Y[:, :nx, :nx])
    dobj = np.trace(X1, axis1=1, axis2=2).sum() + 1e-8 * x0
    # Lagrangian at any primal-feasible (tau, alpha, P):  tau >= dobj + tau r_tau + alpha r_alpha + <r_P, P>
    slack = 2.0 * (abs(r_tau) * kappa + abs(r_alpha) * alpha + np.sqrt(np.sum(r_P ** 2)) * np.sqrt(np.sum(Pbar ** 2)))
    return kappa, dobj, slack, (2 * p * n + 1) * dual['mu_target'][b]


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(0, 3, 3, 3, 2), (13, 2, 30, 4, 1), (11, 2, 6, 12, 4), (12, 2, 4, 24, 8), (777, 2, 64, 24, 8)])
def test_dual_certificate(hc, seed, nb, p, nx, mb):
    """kappa* is pinned from both sides without the oracle and without trusting the solver: the returned primal point is feasible
    with value kappa (upper bound), the exported dual iterate gives a lower bound, and the two differ by the complementarity gap
    N * mu_target the solver stops at (relative gap N * tol <= 1.3e-4 at the benchmark shape)."""
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = hc(p, nx, mb)
    out = h.convexify_batch(A, B, H)
    dual = h.dual(nb)
    for b in range(nb):
        if out['info'][b, 13] != 0.0:
            continue                                  # already convex: no SDP was solved
        assert int(out['status'][b]) == 0
        kappa, dobj, slack, gap = _certificate(A[b], B[b], H[b], out, dual, b)
        assert dobj - slack <= kappa                  # weak duality: lower bound <= attained value
        assert kappa - dobj <= 1.05 * gap + slack     # ... and they are as close as the barrier parameter says
        assert gap / kappa <= (2 * p * (nx + mb) + 1) * 2.0 ** -25 * 1.5
        assert abs(dual['tau'][b] - kappa) <= 1e-12 * kappa
        # the VALUE of the certified gap: kappa* lies in [dobj - slack, kappa]; relative width = N tol (1.2e-4 at p = 64, n = 32) -- the
        # distance of the returned kappa to the optimum that holds whatever solver produced the point (the reference's solver stops at ~1e-8)
        width = (kappa - (dobj - slack)) / kappa
        print(f'certified relative gap on kappa, p={p} n={nx + mb} member {b}: {width:.3e}  (N tol = {(2 * p * (nx + mb) + 1) * 2.0 ** -25:.3e})')
        assert width <= 1.5 * (2 * p * (nx + mb) + 1) * 2.0 ** -25      # (mu_t is tol * kappa rounded to a power of two: up to sqrt(2) above it)


# ----------------------------------------------------------------------------- producer row: sensitivities -> batched inputs
@pytest.mark.parametrize('nb,p,nx,mb,nh', [(2, 3, 3, 2, 5), (3, 40, 9, 6, 7), (1, 4, 24, 8, 70), (2, 5, 2, 2, 1)])
def test_producer_packing_on_device(hc, nb, p, nx, mb, nh):
    """tmpc_pack_sensitivities_host vs the host mirror of pocp.py:322-361 (tunempc_amd.pocp): active rows in order (threshold :73), ragged
    counts incl. stages with no active row, q = -mu' C, stage blocks of the Lagrangian Hessian; exact (pure selection / one dot product)."""
    from tunempc_amd import pocp
    n = nx + mb
    rng = np.random.default_rng(nb * 100 + p + nh)
    C = rng.standard_normal((nb, p, nh, n)); mu = rng.standard_normal((nb, p, nh))
    mu[rng.random((nb, p, nh)) < 0.6] = 0.0                      # inactive constraints: exact zero multipliers
    mu[0, 0] = 0.0                                               # a stage without any active constraint -> None in the reference
    mu[-1, -1, :] = 1e-16                                        # below the threshold 1e-15
    Hbig = rng.standard_normal((nb, p * n, p * n))
    h = hc(p, nx, mb)
    out = pocp.pack_batch_device(h, C=C, mu=mu, Hbig=Hbig)
    for b in range(nb):
        C_As, idx = pocp.active_set(list(C[b]), list(mu[b]))
        qs = pocp.cost_gradient(list(mu[b]), list(C[b]))
        Hs = pocp.stage_hessians(Hbig[b], n, p)
        for k in range(p):
            nc = 0 if C_As[k] is None else C_As[k].shape[0]
            assert int(out['nc'][b, k]) == nc
            if nc:
                assert np.array_equal(out['C_As'][b, k, :nc], C_As[k]) and list(out['idx'][b, k, :nc]) == idx[k]
            assert not out['C_As'][b, k, nc:].any() and (out['idx'][b, k, nc:] == -1).all()
            assert np.abs(out['q'][b, k] - qs[k].ravel()).max() <= 1e-14 * max(1.0, np.abs(qs[k]).max())
            assert np.array_equal(out['H'][b, k], Hs[k])
    assert (out['nc'][0, 0] == 0) and (out['nc'][-1, -1] == 0)
    # no path constraints: q = zeros (pocp.py:361)
    o2 = pocp.pack_batch_device(h, Hbig=Hbig)
    assert not o2['q'].any() and np.array_equal(o2['H'], out['H'])


# ----------------------------------------------------------------------------- consumer row: tracking reference
@pytest.mark.parametrize('n,ns', [(4, 1), (5, 30), (15, 40), (32, 257)])
def test_tracking_reference_parity(hc, n, ns):
    """tmpc_tracking_reference_host vs the numpy restatement of pmpc.py:961-974 (W = Hc/ts, yref = wref - Hc^-1 q)."""
    from oracle import tracking_oracle
    h = hc(1, n, 0)
    rng = np.random.default_rng(n * 100 + ns)
    X = rng.standard_normal((ns, n, n)); H = X @ np.swapaxes(X, 1, 2) / n + 0.3 * np.eye(n)
    q = rng.standard_normal((ns, n)); w = rng.standard_normal((ns, n))
    W, y, info = h.tracking_reference(H, q, w, 0.05)
    Wr, yr = tracking_oracle.tracking_reference(H, q, w, 0.05)
    assert not info.any()
    assert rel(W, Wr) < 1e-15 and rel(y, yr) < 1e-11
    Hbad = H.copy(); Hbad[0] = -Hbad[0]
    _, _, info = h.tracking_reference(Hbad, q, w, 0.05)
    assert info[0] > 0 and not info[1:].any()


def test_tracking_reference_python_mirror(hc):
    """tunempc_amd.pmpc.tracking_reference over lists (the shapes Pmpc holds: q rows 1 x n, wref columns n x 1) after a real convexify."""
    from tunempc_amd import pmpc, convexifier
    from oracle import convexify_oracle as orc, tracking_oracle
    A, B, H = orc.gen_problem(20, 6, 3, 2)[:3]
    Q = [H[k][:3, :3] for k in range(6)]; R = [H[k][3:, 3:] for k in range(6)]; N = [H[k][:3, 3:] for k in range(6)]
    dHc, _, _, _ = convexifier.convexify(list(A), list(B), Q, R, N, opts={'rho': 1e-3, 'solver': 'hip', 'force': False})
    Hc = [H[k] + dHc[k] for k in range(6)]
    rng = np.random.default_rng(1)
    q = [rng.standard_normal((1, 5)) for _ in range(6)]; w = [rng.standard_normal((5, 1)) for _ in range(6)]
    W, y = pmpc.tracking_reference(Hc, q, w, 0.2)
    Wr, yr = tracking_oracle.tracking_reference(np.stack(Hc), np.stack([v[0] for v in q]), np.stack([v[:, 0] for v in w]), 0.2)
    assert rel(np.stack(W), Wr) < 1e-14 and rel(np.stack(y), yr) < 1e-10
    with pytest.raises(ValueError, match='not positive definite'):
        pmpc.tracking_reference([-m for m in Hc], q, w, 0.2)


def test_supplement_with_constraint_and_regularisation_terms(hc):
    """convexHessianSuppl with the G/Fg, ragged C/F and T terms (convexifier.py:196-204) vs the numpy restatement."""
    from tunempc_amd import convexifier
    from oracle import convexify_oracle as orc
    rng = np.random.default_rng(11)
    p, nx, mb, ng = 5, 4, 2, 3
    n = nx + mb
    A = rng.standard_normal((p, nx, nx)); B = rng.standard_normal((p, nx, mb))
    P = rng.standard_normal((p, nx, nx)); P = P + np.swapaxes(P, 1, 2)
    G = [rng.standard_normal((ng, n)) for _ in range(p)]; Fg = [rng.uniform(0, 1, (ng, 1)) for _ in range(p)]
    C = [rng.standard_normal((2, n)), None, rng.standard_normal((1, n)), rng.standard_normal((3, n)), None]
    F = [rng.uniform(0, 1, (2, 1)), None, rng.uniform(0, 1, (1, 1)), rng.uniform(0, 1, (3, 1)), None]
    T = [rng.uniform(0, 1, (n, n)) for _ in range(p)]
    Q = R = N = None
    for kw in [dict(), dict(G=G, Fg=Fg), dict(C=C, F=F), dict(T=T), dict(G=G, Fg=Fg, C=C, F=F, T=T), dict(C=C)]:
        dHc, dQc, dRc, dNc = convexifier.convexHessianSuppl(list(A), list(B), Q, R, N, list(P), **kw)
        ref = orc.convex_hessian_suppl(A, B, P, **kw)
        assert rel(np.stack(dHc), ref[0]) < 1e-14, kw.keys()
        assert rel(np.stack(dQc), ref[1]) < 1e-14 and rel(np.stack(dRc), ref[2]) < 1e-14 and rel(np.stack(dNc), ref[3]) < 1e-14
        assert all(np.array_equal(d, d.T) for d in dHc)


@pytest.mark.parametrize('variant', [dict(flags=32), dict(tuning=dict(fuse_fwd=0, chord_step=0)), dict(tuning=dict(small_blocks=0)), dict(tuning=dict(eig_pretest=0)),
                                     dict(tuning=dict(graph=0)), dict(lanes=1), dict(lanes=3), dict(flags=1), dict(tuning=dict(persistent=2)), dict(tuning=dict(persistent=0))], ids=str)
def test_kernel_variants_behind_handle_options(golden_dir, variant):
    """The kernel variants that rounds 1-3 hid behind environment variables, now options of the handle (tmpc_set_tuning, tmpc_create_ex) or debug flags
    (tunempc_hip_debug.h): register-staged factorisation kernels (flag 32: the path of blocks wider than 320), separate forward sweep without chord steps, the batched
    launch sequence for 16-wide blocks, every step-length eigenvalue computed, one / three lanes, scalar-FMA GEMM fragments, the persistent one-launch loop of
    tmpc_persist.h forced on / off (round 5; it serves the two small shapes, the others fall through) -- each against four golden vectors."""
    from tunempc_amd._lib import HipConvexifier
    worst = 0.0
    for name in ('c2_unicycle_shape', 'c3_evaporation_shape', 'mid_n16', 'awe_shape_n15'):
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        A, B, H = g['A'], g['B'], g['H']
        h = HipConvexifier(A.shape[1], A.shape[2], B.shape[3], flags=variant.get('flags', 0), lanes=variant.get('lanes', 0))
        if 'tuning' in variant:
            h.set_tuning(**variant['tuning'])
        out = h.convexify_batch(A, B, H)
        h.close()
        for b in range(A.shape[0]):
            assert int(out['status'][b]) == int(g['status'][b]), (name, b, out['status'][b], g['status'][b])
            worst = max(worst, np.linalg.norm(out['Hc'][b] - g['Hc'][b]) / np.linalg.norm(g['Hc'][b]))
    assert worst < PARITY, (variant, worst)


# ----------------------------------------------------------------------------- stage blocks wider than 32
@pytest.mark.parametrize('seed,nb,p,nx,mb', [(201, 2, 4, 30, 10), (202, 2, 6, 36, 12), (203, 1, 3, 40, 24), (204, 2, 8, 33, 1), (205, 2, 1, 34, 4), (206, 2, 2, 35, 3), (209, 1, 3, 48, 8),
                                             (210, 1, 2, 40, 32), (211, 1, 3, 60, 12), (212, 1, 1, 48, 48), (213, 2, 2, 24, 56)])      # round 5: n = 72, 72 (blocks of 1830), 96, 80
def test_large_stage_blocks_parity(seed, nb, p, nx, mb):
    """32 < n = nx + m <= 96 (the reference accepts any size: preprocessing.py:157-185; above 64 the plain model only): the generic per-stage kernels of csrc/tmpc_big.h with the
    register-staged block factorisation (blocks up to 820 wide here) against the C++ CPU port, plain Step 1 model, to the 1e-8 bar; the structural
    invariants; p = 1 and p = 2 included."""
    from tunempc_amd._lib import HipConvexifier
    import cpu_ipm
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=nb)
    out = h.convexify_batch(A, B, H)
    h.close()
    ref = cpu_ipm.convexify_batch(A, B, H, threads=8)
    for b in range(nb):
        assert int(out['status'][b]) == 0 == int(ref['status'][b])
        assert rel(out['Hc'][b], ref['Hc'][b]) < PARITY, (b, rel(out['Hc'][b], ref['Hc'][b]))
        assert abs(out['kappa'][b] - ref['kappa'][b]) < 1e-9 * ref['kappa'][b]
        ev = np.linalg.eigvalsh(out['Hc'][b])
        assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= out['kappa'][b] * (1 + 1e-9)
        dH = co.convex_hessian_suppl(A[b], B[b], out['P'][b])[0]
        assert rel(out['Hc'][b] - H[b], dH) < 1e-10


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(3, 2, 6, 4, 2), (70, 3, 16, 12, 4), (95, 2, 5, 24, 8), (61, 2, 1, 5, 2)])
def test_generic_stage_kernels_match_the_tuned_ones(hc, seed, nb, p, nx, mb):
    """TMPC_DEBUG_FLAG_GENERIC_STAGE (64): the generic per-stage kernels at n <= 32, against the tuned kernels on the same inputs -- same iteration
    counts, Hc to 1e-9 (they differ in rounding only: one thread per output entry and a Jacobi eigenvalue iteration instead of MFMA tiles, Householder + Sturm)."""
    from tunempc_amd._lib import HipConvexifier
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    # (both in fp64 throughout: with the single-precision updates of round 6 on, whether a pivot freezes under them -- and the iteration is repeated -- depends on
    # the rounding of the stage kernels, which is exactly what differs between the two sets; tests/test_gpu_lowp.py covers that feature)
    h0 = HipConvexifier(p, nx, mb, chunk=nb); h0.set_tuning(lowp_switch=0.0)
    ref = h0.convexify_batch(A, B, H)
    h0.close()
    h = HipConvexifier(p, nx, mb, chunk=nb, flags=64); h.set_tuning(lowp_switch=0.0)
    out = h.convexify_batch(A, B, H)
    h.close()
    assert np.array_equal(out['status'], ref['status']) and np.array_equal(out['iters'], ref['iters'])
    for b in range(nb):
        assert rel(out['Hc'][b], ref['Hc'][b]) < 1e-9, (b, rel(out['Hc'][b], ref['Hc'][b]))


def test_dropin_convexify_large_block():
    """convexifier.convexify at nx + nu = 38 (rounds 1-3 raised NotImplementedError): same call as the reference's, lists in, lists out."""
    from tunempc_amd import convexifier
    p, nx, mb = 3, 30, 8
    A, B, H = co.gen_batch(207, 1, p, nx, mb)
    Q = [H[0, k][:nx, :nx] for k in range(p)]; R = [H[0, k][nx:, nx:] for k in range(p)]; N = [H[0, k][:nx, nx:] for k in range(p)]
    dHc, dQc, dRc, dNc = convexifier.convexify([A[0, k] for k in range(p)], [B[0, k] for k in range(p)], Q, R, N, opts={'rho': 1e-3, 'solver': 'hip', 'force': False})
    r = co.convexify_arrays(A[0], B[0], H[0])
    assert rel(np.stack(dHc), r['dHc']) < 1e-7 and all(np.linalg.eigvalsh(H[0, k] + dHc[k]).min() > 0 for k in range(p))
    assert np.array_equal(dQc[1], dHc[1][:nx, :nx]) and np.array_equal(dNc[2], dHc[2][:nx, nx:])
    convexifier.release_handles()


# ----------------------------------------------------------------------------- the dual certificate for the models with stage-local multipliers
@pytest.mark.parametrize('model', ['G', 'step2'])
def test_dual_certificate_with_multipliers(model):
    """As test_dual_certificate, for Step 1 with G (cost-free multipliers phi >= 0, duals z) and for the Step 2 model (the norm terms rho ||F_k||, rho ||Fg_k|| as
    epigraph variables t_e with arrow LMIs, primal blocks X_e): numpy only, no oracle.  The exported dual iterate (tmpc_get_dual_host + tmpc_get_dual_con_host) is
    feasible up to tiny residuals and its objective is a lower bound of the optimal value (weak duality); the attained primal value kappa (+ sum t_e) is an upper
    bound; they differ by the complementarity gap N mu_target."""
    from tunempc_amd._lib import HipConvexifier
    p, nx, mb, nb, ng, nc, rho = 5, 5, 2, 2, 2, 3, 1e-2
    n = nx + mb
    A, B, H = co.gen_batch(8100, nb, p, nx, mb)
    rng = np.random.default_rng(81)
    G = rng.standard_normal((nb, p, ng, n)); Cc = rng.standard_normal((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            Cc[b, k, ncnt[b, k]:] = 0.0
    if model == 'G':
        h = HipConvexifier(p, nx, mb, ng=ng, chunk=nb)
        out = h.convexify_eq_batch(A, B, H, G)
        J = G; rows = np.full((nb, p), ng)
    else:
        h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb)
        J = np.concatenate([G, Cc], axis=2); rows = ng + ncnt
        out = h.convexify_step2_batch(A, B, H, J, ncnt, rho)
    dual = h.dual(nb); dc = h.dual_con(nb, arrows=(model == 'step2'))
    h.close()
    for b in range(nb):
        assert int(out['status'][b]) == 0
        primal, dobj, slack, gap = _certificate_con(A[b], B[b], H[b], J[b], rows[b], ncnt[b], ng, rho, out, dual, dc, b, model == 'step2')
        assert dobj - slack <= primal
        assert primal - dobj <= 1.05 * gap + slack, (model, b, primal - dobj, gap, slack)
        print(f'{model}: certified relative gap {(primal - (dobj - slack)) / primal:.3e} (N mu_t / value = {gap / primal:.3e}, residual slack {slack / primal:.1e})')


def _certificate_con(A, B, H, J, rows, ncnt, ng, rho, out, dual, dc, b, arrows):
    """Certificate of member b of a model with stage-local multipliers, numpy only: (attained primal value, dual objective, slack from the dual residuals,
    N * mu_target).  A, B, H, J [p, ...] of the member; rows [p] = rows of [G_k; C_k] present; arrows: Step 2 (norm terms)."""
    p, nx, _ = A.shape
    n = H.shape[1]
    s = out['info'][b, 0]; sbeta = out['info'][b, 1]
    alpha, kappa = out['alpha'][b], out['kappa'][b]
    Hb = s * 0.5 * (H + H.transpose(0, 2, 1))
    V = np.concatenate([A, B], axis=2)
    Pbar = out['P'][b] * (s * alpha)
    phi, z = dc['phi'][b], dc['z'][b]
    M = alpha * Hb + V.transpose(0, 2, 1) @ np.roll(Pbar, -1, axis=0) @ V
    M[:, :nx, :nx] -= Pbar
    ncone = 2 * p * n + 1
    for k in range(p):
        for i in range(rows[k]):
            M[k] += phi[k, i] * np.outer(J[k, i], J[k, i])
            assert phi[k, i] > 0 and z[k, i] > 0
            ncone += 1
    ev = np.linalg.eigvalsh(M)
    assert ev.min() >= 1.0 - 1e-9 and ev.max() <= kappa * (1 + 1e-9) and alpha > 1e-8                     # primal feasible
    X1, X2, x0 = dual['X1'][b], dual['X2'][b], dual['x0'][b]
    assert np.linalg.eigvalsh(X1).min() > 0 and np.linalg.eigvalsh(X2).min() > 0 and x0 > 0
    Y = X1 - X2
    r_tau = 1.0 - np.trace(X2, axis1=1, axis2=2).sum()
    r_alpha = -np.sum(Hb * Y) - x0
    W = V @ Y @ V.transpose(0, 2, 1)
    r_P = -(np.roll(W, 1, axis=0) - Y[:, :nx, :nx])
    primal = kappa
    slack = 2.0 * (abs(r_tau) * kappa + abs(r_alpha) * alpha + np.sqrt(np.sum(r_P ** 2)) * np.sqrt(np.sum(Pbar ** 2)))
    wr = rho * sbeta / s
    for k in range(p):
        r_phi = np.array([-J[k, i] @ Y[k] @ J[k, i] - z[k, i] for i in range(rows[k])])
        if arrows:
            e = 0
            for (a0, m) in ([(0, ng)] if ng else []) + ([(ng, int(ncnt[k]))] if ncnt[k] else []):
                Xe = dc['aX'][b, k, e][:m + 1, :m + 1]; te = dc['at'][b, k, e]
                assert np.linalg.eigvalsh(Xe).min() > 0
                assert te >= wr * np.linalg.norm(phi[k, a0:a0 + m]) * (1 - 1e-12)                        # the arrow LMI of the epigraph holds
                r_phi[a0:a0 + m] -= 2.0 * wr * Xe[0, 1:]
                slack += 2.0 * abs(1.0 - np.trace(Xe)) * te
                primal += te
                ncone += m + 1
                e += 1
        slack += 2.0 * np.sum(np.abs(r_phi) * phi[k, :rows[k]])
    dobj = np.trace(X1, axis1=1, axis2=2).sum() + 1e-8 * x0
    return primal, dobj, slack, ncone * dual['mu_target'][b]


def test_large_block_side_entries():
    """the other entry points of the boundary at 32 < n <= 64 (generic kernels): eigen-scan, supplement of P, tracking reference -- against numpy"""
    from tunempc_amd._lib import HipConvexifier
    p, nx, mb, nb = 3, 28, 10, 2
    n = nx + mb
    A, B, H = co.gen_batch(208, nb, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=nb)
    sc = h.eig_scan(H)
    ev = np.linalg.eigvalsh(H)
    assert np.abs(sc[..., 0] - ev[..., 0]).max() < 1e-11 * np.abs(ev).max() and np.abs(sc[..., 1] - ev[..., -1]).max() < 1e-11 * np.abs(ev).max()
    assert np.abs(sc[..., 2] - np.abs(ev).min(-1)).max() < 1e-11 * np.abs(ev).max() and np.abs(sc[..., 3] - np.abs(ev).max(-1)).max() < 1e-11 * np.abs(ev).max()
    rng = np.random.default_rng(2)
    P = rng.standard_normal((nb, p, nx, nx)); P = P + P.transpose(0, 1, 3, 2)
    dH = h.supplement_batch(A, B, P)
    for b in range(nb):
        assert rel(dH[b], co.convex_hessian_suppl(A[b], B[b], P[b])[0]) < 1e-13
    out = h.convexify_batch(A, B, H)
    q = rng.standard_normal((nb * p, n)); wref = rng.standard_normal((nb * p, n))
    W, yref = h.tracking_reference(out['Hc'].reshape(nb * p, n, n), q, wref, 0.1)[:2]
    Hs = out['Hc'].reshape(nb * p, n, n)
    assert rel(W, Hs / 0.1) < 1e-14
    assert rel(yref, wref - np.linalg.solve(Hs, q[..., None])[..., 0]) < 1e-10
    h.close()


# ----------------------------------------------------------------------------- multipliers of G / C at 32 < n <= 64 (second half)
def _mult_model(seed, nb, p, nx, mb, ng, nc):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 1)
    n = nx + mb
    G = rng.standard_normal((nb, p, ng, n)); C = np.zeros((nb, p, nc, n))
    ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            C[b, k, :ncnt[b, k]] = rng.standard_normal((ncnt[b, k], n))
    return A, B, H, G, C, ncnt


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng,nc', [(11, 2, 5, 5, 2, 2, 3), (12, 2, 8, 12, 4, 3, 4), (13, 1, 4, 24, 8, 4, 5), (14, 2, 1, 6, 2, 1, 2), (15, 2, 6, 10, 3, 0, 4)])
def test_generic_multiplier_kernels_match_the_tuned_ones(seed, nb, p, nx, mb, ng, nc):
    """debug flag 64 with G / C rows: the <true> forms of k_phi_pre / k_phi_rhs / k_phi_dir (matrices read from global memory, 64-long vectors) on the generic
    per-stage kernels, against the tuned kernels on the same inputs -- Step 1 with G and the Step 2 model; same iteration counts, Hc and multipliers to 1e-9"""
    from tunempc_amd._lib import HipConvexifier
    A, B, H, G, C, ncnt = _mult_model(seed, nb, p, nx, mb, ng, nc)
    J = np.concatenate([G, C], axis=2)
    res = []
    for flags in (0, 64):
        h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=nb, flags=flags)
        h.set_tuning(lowp_switch=0.0)          # (fp64 throughout: the two kernel sets are compared for rounding only, see test_generic_stage_kernels_match_the_tuned_ones)
        o2 = h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
        o1 = h.convexify_eq_batch(A, B, H, G) if ng else None
        h.close()
        res.append((o1, o2))
    for a, g in zip(res[0], res[1]):
        if a is None:
            continue
        assert np.array_equal(a['status'], g['status']) and np.array_equal(a['iters'], g['iters'])
        for b in range(nb):
            assert rel(g['Hc'][b], a['Hc'][b]) < 1e-9
        fa = a['FgF'] if 'FgF' in a else a['Fg']; fg = g['FgF'] if 'FgF' in g else g['Fg']
        assert np.abs(fa - fg).max() < 1e-9 * max(np.abs(fa).max(), 1e-3)


@pytest.mark.parametrize('seed,p,nx,mb,ng,nc', [(21, 3, 24, 10, 2, 3), (22, 4, 20, 16, 3, 2), (23, 2, 30, 12, 0, 4), (24, 3, 26, 8, 2, 0), (25, 2, 40, 8, 3, 3),
                                                (31, 3, 8, 4, 20, 18), (32, 2, 20, 10, 24, 24), (33, 3, 12, 6, 31, 0), (34, 2, 36, 12, 24, 24), (35, 2, 6, 2, 0, 31)])
def test_large_stage_blocks_with_multipliers(seed, p, nx, mb, ng, nc):
    """Steps 1 and 2 at 32 < n <= 64 and / or with more than 16 rows of G_k / C_k per stage (up to 31 each; n = 48 with 24 + 24 rows is the case the
    round-3 review names) against the numpy oracle: Hc (= H + the supplement with the multiplier terms), kappa"""
    from tunempc_amd._lib import HipConvexifier
    A, B, H, G, C, ncnt = _mult_model(seed, 1, p, nx, mb, ng, nc)
    Cl = [C[0, k, :ncnt[0, k]] if ncnt[0, k] else None for k in range(p)]
    r = co.sdp_step1(A[0], B[0], H[0], G=G[0] if ng else None, C=Cl if nc else None, rho=1e-2 if nc else None)
    dH = co.convex_hessian_suppl(A[0], B[0], r['P'], G=G[0] if ng else None, Fg=r.get('Fg'), C=Cl if nc else None, F=r.get('F'))[0]
    h = HipConvexifier(p, nx, mb, ng=ng, nc=nc, chunk=1)
    o = h.convexify_step2_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 1e-2) if nc else h.convexify_eq_batch(A, B, H, G)
    h.close()
    assert int(o['status'][0]) == 0 and r['ipm_status'] == 'optimal'
    assert rel(o['Hc'][0], H[0] + dH) < PARITY
    assert abs(o['kappa'][0] - r['kappa']) < 1e-9 * r['kappa']
    assert np.linalg.eigvalsh(o['Hc'][0]).min() > 0


def test_large_block_supplement_with_multiplier_terms():
    """tmpc_supplement_terms_batch_host at n = 40 with the J' diag(w) J terms (generic kernel) against the oracle's convexHessianSuppl"""
    from tunempc_amd._lib import HipConvexifier
    p, nx, mb, nb, nr = 3, 30, 10, 2, 4
    A, B, H = co.gen_batch(210, nb, p, nx, mb)
    rng = np.random.default_rng(3)
    P = rng.standard_normal((nb, p, nx, nx)); P = P + P.transpose(0, 1, 3, 2)
    J = rng.standard_normal((nb, p, nr, nx + mb)); wts = rng.uniform(0.0, 1.0, (nb, p, nr))
    h = HipConvexifier(p, nx, mb, chunk=nb)
    dH = h.supplement_terms_batch(A, B, P, J=J, wts=wts)
    T = rng.uniform(0.1, 1.0, (nb, p, nx + mb, nx + mb)); T = T + T.transpose(0, 1, 3, 2)      # round 5: the T term of Step 3 at n > 32 as well
    dHT = h.supplement_terms_batch(A, B, P, J=J, wts=wts, T=T)
    dHT0 = h.supplement_terms_batch(A, B, P, T=T)
    h.close()
    for b in range(nb):
        want = co.convex_hessian_suppl(A[b], B[b], P[b], G=J[b], Fg=wts[b])[0]
        assert rel(dH[b], want) < 1e-13
        assert rel(dHT[b], co.convex_hessian_suppl(A[b], B[b], P[b], G=J[b], Fg=wts[b], T=T[b])[0]) < 1e-13
        assert rel(dHT0[b], co.convex_hessian_suppl(A[b], B[b], P[b], T=T[b])[0]) < 1e-13


def test_many_row_handle_matches_the_small_one():
    """a handle with room for 3 + 31 rows per stage keeps the per-row vectors and Gram products of k_phi_pre in global memory (its <.., true> form);
    on inputs with at most 4 rows of C_k it must return what the handle with room for 3 + 4 rows (LDS form) returns: same iterations, Hc to 1e-10
    (the blocks of the two handles are padded differently, so the factorisations round differently)"""
    from tunempc_amd._lib import HipConvexifier
    nb, p, nx, mb, ng, nc = 2, 5, 6, 3, 3, 4
    A, B, H, G, C, ncnt = _mult_model(41, nb, p, nx, mb, ng, nc)
    outs = []
    for room in (nc, 31):
        Cp = np.zeros((nb, p, room, nx + mb)); Cp[:, :, :nc] = C
        h = HipConvexifier(p, nx, mb, ng=ng, nc=room, chunk=nb)
        h.set_tuning(lowp_switch=0.0)
        outs.append(h.convexify_step2_batch(A, B, H, np.concatenate([G, Cp], axis=2), ncnt, 1e-2))
        h.close()
    a, g = outs
    assert np.array_equal(a['status'], g['status']) and np.array_equal(a['iters'], g['iters']) and (a['status'] == 0).all()
    assert rel(g['Hc'], a['Hc']) < 1e-10
    assert np.abs(g['FgF'][:, :, :ng + nc] - a['FgF']).max() < 1e-10 * max(1.0, np.abs(a['FgF']).max())


def test_dropin_convexify_large_block_with_constraints():
    """convexifier.convexify(..., G=, C=) at nx + nu = 36 with 20 active rows at one stage: the same call as the reference's; Step 1 (with G) is feasible here, so the
    result is that of the oracle's Step 1 with G"""
    from tunempc_amd import convexifier
    p, nx, mb, ng = 3, 26, 10, 2
    n = nx + mb
    A, B, H = co.gen_batch(211, 1, p, nx, mb)
    rng = np.random.default_rng(212)
    G = [rng.standard_normal((ng, n)) for _ in range(p)]
    C = [rng.standard_normal((20, n)), None, rng.standard_normal((3, n))]
    Q = [H[0, k][:nx, :nx] for k in range(p)]; R = [H[0, k][nx:, nx:] for k in range(p)]; N = [H[0, k][:nx, nx:] for k in range(p)]
    dHc = convexifier.convexify([A[0, k] for k in range(p)], [B[0, k] for k in range(p)], Q, R, N, G=G, C=C, opts={'rho': 1e-3, 'solver': 'hip', 'force': False})[0]
    r = co.sdp_step1(A[0], B[0], H[0], G=np.stack(G))
    want = co.convex_hessian_suppl(A[0], B[0], r['P'], G=np.stack(G), Fg=r['Fg'])[0]
    assert r['ipm_status'] == 'optimal' and rel(np.stack(dHc), want) < 1e-7
    assert all(np.linalg.eigvalsh(H[0, k] + dHc[k]).min() > 0 for k in range(p))
    # the Step 2 model of the same data through the batched mirror (what convexify runs when Step 1 is infeasible)
    Cp = np.zeros((1, p, 20, n)); ncnt = np.zeros((1, p), np.int32)
    for k in range(p):
        if C[k] is not None:
            Cp[0, k, :C[k].shape[0]] = C[k]; ncnt[0, k] = C[k].shape[0]
    o2 = convexifier.convexify_step2_batch(A, B, H, Cp, ncnt, 1e-3, G=np.stack(G)[None])
    r2 = co.sdp_step1(A[0], B[0], H[0], G=np.stack(G), C=C, rho=1e-3)
    want2 = co.convex_hessian_suppl(A[0], B[0], r2['P'], G=np.stack(G), Fg=r2['Fg'], C=C, F=r2['F'])[0]
    assert int(o2['status'][0]) == 0 and rel(o2['dHc'][0], want2) < 1e-7
    convexifier.release_handles()


# ----------------------------------------------------------------------------- Step 3 on the generic per-stage kernels
@pytest.mark.parametrize('seed,nb,p,nx,mb', [(1, 2, 3, 3, 2), (2, 2, 4, 6, 2), (3, 1, 2, 12, 4)])
def test_step3_generic_stage_kernels_match_the_tuned_ones(seed, nb, p, nx, mb):
    """debug flag 64 with Step 3: k_t3_schur<true> (X_r, S_r^-1 read from global memory, X_r V' / S_r^-1 V' in the stage scratch) and the T_k terms of the generic
    per-stage kernels against the tuned kernels -- same iteration counts, Hc and T to 1e-9; also with rows of G / C in the same solve"""
    from tunempc_amd._lib import HipConvexifier
    A, B, H = co.gen_batch(700 + seed, nb, p, nx, mb)
    res = []
    for flags in (0, 64):
        h = HipConvexifier(p, nx, mb, chunk=nb, step3=True, flags=flags)
        res.append(h.convexify_step3_batch(A, B, H, 1e-2))
        h.close()
    a, g = res
    assert np.array_equal(a['status'], g['status']) and np.array_equal(a['iters'], g['iters']) and (a['iters'] > 0).all()
    assert rel(g['Hc'], a['Hc']) < 1e-9 and rel(g['T'], a['T']) < 1e-9
    ng, nc = 2, 3
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((nb, p, ng + nc, nx + mb)); ncnt = rng.integers(0, nc + 1, size=(nb, p)).astype(np.int32)
    for b in range(nb):
        for k in range(p):
            J[b, k, ng + ncnt[b, k]:] = 0.0
    res = []
    for flags in (0, 64):
        h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc, step3=True, flags=flags)
        res.append(h.convexify_step3_con_batch(A, B, H, J, ncnt, 1e-2))
        h.close()
    a, g = res
    assert np.array_equal(a['status'], g['status']) and np.array_equal(a['iters'], g['iters'])
    assert rel(g['Hc'], a['Hc']) < 1e-9 and rel(g['T'], a['T']) < 1e-9 and np.abs(g['FgF'] - a['FgF']).max() < 1e-9 * max(1.0, np.abs(a['FgF']).max())


def test_step3_large_stage_block():
    """Step 3 at nx + nu = 34 (blocks of 300 + 595 + 1 = 896) against the numpy oracle: Hc, T (every entry > 0), kappa"""
    from tunempc_amd._lib import HipConvexifier
    p, nx, mb = 2, 24, 10
    A, B, H = co.gen_batch(300, 1, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=1, step3=True)
    o = h.convexify_step3_batch(A, B, H, 1e-2)
    h.close()
    r = co.sdp_step1(A[0], B[0], H[0], rho=1e-2, force=True)
    dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], T=r['T'])[0]
    assert int(o['status'][0]) == 0 and r['ipm_status'] == 'optimal'
    assert rel(o['Hc'][0], H[0] + dHc) < PARITY and rel(o['T'][0], r['T']) < PARITY and (o['T'][0] > 0).all()
    assert abs(o['kappa'][0] - r['kappa']) < 1e-9 * r['kappa']


# ----------------------------------------------------------------------------- Schur blocks beyond 1552 (compact LDS image of the substitution kernels)
@pytest.mark.parametrize('seed,p,nx,mb', [(401, 2, 60, 4), (402, 3, 63, 1)])
def test_widest_plain_blocks(seed, p, nx, mb):
    """the plain model at nx = 60 and nx = 63 (n = 64; Schur blocks of 1830 and 2016) against the C++ port"""
    from tunempc_amd._lib import HipConvexifier
    import cpu_ipm
    A, B, H = co.gen_batch(seed, 1, p, nx, mb)
    h = HipConvexifier(p, nx, mb, chunk=1)
    o = h.convexify_batch(A, B, H)
    h.close()
    c = cpu_ipm.convexify_batch(A, B, H, threads=8)
    assert int(o['status'][0]) == 0 == int(c['status'][0])
    assert rel(o['Hc'][0], c['Hc'][0]) < PARITY and abs(o['kappa'][0] - c['kappa'][0]) < 1e-9 * c['kappa'][0]


@pytest.mark.parametrize('p,nx,mb,ng,nc', [(2, 30, 10, 0, 0), (2, 36, 12, 24, 24), (1, 40, 24, 0, 0)])
def test_step3_at_the_sizes_the_review_names(p, nx, mb, ng, nc):
    """Step 3 at n = 40 (nx = 30, m = 10) and at n = 48 with 24 + 24 rows of G_k / C_k in the same solve (Schur blocks of 1286 and 1893) against the numpy oracle
    (the second takes the oracle ~90 s on the GPU box's host cores); round 5: n = 64 with nx = 40 -- blocks of 2901, beyond the 2384 the substitution kernels held
    in LDS until round 4 (review of round 4, item 4 iii)"""
    from tunempc_amd._lib import HipConvexifier
    A, B, H, G, C, ncnt = _mult_model(34, 1, p, nx, mb, ng, nc)
    h = HipConvexifier(p, nx, mb, chunk=1, ng=ng, nc=nc, step3=True)
    if ng or nc:
        o = h.convexify_step3_con_batch(A, B, H, np.concatenate([G, C], axis=2), ncnt, 1e-2)
        Cl = [C[0, k, :ncnt[0, k]] if ncnt[0, k] else None for k in range(p)]
        r = co.sdp_step1(A[0], B[0], H[0], G=G[0], C=Cl, rho=1e-2, force=True)
        dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], G=G[0], Fg=r['Fg'], C=Cl, F=r['F'], T=r['T'])[0]
    else:
        o = h.convexify_step3_batch(A, B, H, 1e-2)
        r = co.sdp_step1(A[0], B[0], H[0], rho=1e-2, force=True)
        dHc = co.convex_hessian_suppl(A[0], B[0], r['P'], T=r['T'])[0]
    h.close()
    assert int(o['status'][0]) == 0 and r['ipm_status'] == 'optimal'
    assert rel(o['Hc'][0], H[0] + dHc) < PARITY and rel(o['T'][0], r['T']) < PARITY and (o['T'][0] > 0).all()
    assert abs(o['kappa'][0] - r['kappa']) < 1e-9 * r['kappa']


def test_iteration_graph_replay_is_bit_identical():
    """TMPC_TUNE_GRAPH (default on for problems whose Schur blocks are one tile): the launch sequence of an IPM iteration replayed as a captured hipGraph returns the
    bits of the plain launch sequence -- batch 1, a batch over two lanes with ragged iteration counts, the models with multipliers and Step 3, repeated calls (cache hits)"""
    from tunempc_amd._lib import HipConvexifier
    for (seed, nb, p, nx, mb) in [(200000, 1, 30, 4, 1), (11, 7, 8, 4, 1), (3, 5, 6, 4, 2)]:
        A, B, H = co.gen_batch(seed, nb, p, nx, mb)
        rng = np.random.default_rng(seed)
        J = rng.standard_normal((nb, p, 3, nx + mb)); ncnt = rng.integers(0, 3, size=(nb, p)).astype(np.int32)
        for b in range(nb):
            for k in range(p):
                J[b, k, 1 + ncnt[b, k]:] = 0.0
        outs = []
        for g in (0, 1):
            h = HipConvexifier(p, nx, mb, ng=1, nc=2, chunk=nb); h.set_tuning(graph=g)
            o = [h.convexify_batch(A, B, H), h.convexify_eq_batch(A, B, H, J[:, :, :1]), h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2), h.convexify_batch(A, B, H)]
            h.close()
            h3 = HipConvexifier(p, nx, mb, step3=True, chunk=nb); h3.set_tuning(graph=g)
            o.append(h3.convexify_step3_batch(A, B, H, 1e-2))
            h3.close()
            outs.append(o)
        for a, g in zip(*outs):
            for k in ('Hc', 'P', 'kappa', 'status', 'iters'):
                assert np.array_equal(a[k], g[k]), (seed, k)
        assert np.array_equal(outs[1][0]['Hc'], outs[1][3]['Hc'])


# ----------------------------------------------------------------------------- round 5: the whole interior-point loop of a small problem as one launch (tmpc_persist.h)
@pytest.mark.parametrize('seed,nb,p,nx,mb', [(13, 2, 30, 4, 1), (20, 4, 1, 3, 1), (30, 4, 2, 3, 1), (40, 3, 5, 2, 2), (41, 2, 7, 5, 1), (42, 3, 17, 5, 3), (43, 2, 33, 1, 1), (44, 1, 160, 2, 1),
                                             (45, 2, 16, 4, 4), (46, 1, 48, 3, 5)])
def test_persistent_small_kernel_parity(seed, nb, p, nx, mb):
    """One 16-wave workgroup runs every iteration of a problem (n <= 8, nx <= 5, plain model) -- forced on whatever the batch -- against the CPU oracle (1e-8) and against
    the launch-sequence path (same code, sums taken by one wave instead of four: rounding only, same iteration counts); periods of 1, 2, one round, a round boundary
    (16, 17, 33), the longest the LDS image of the substitution takes (160), early-exit members."""
    from tunempc_amd._lib import HipConvexifier
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    outs = []
    for mode in (2, 0):
        h = HipConvexifier(p, nx, mb, chunk=nb)
        h.set_tuning(persistent=mode)
        outs.append(h.convexify_batch(A, B, H))
        h.close()
    o, s = outs
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b])
        assert int(o['status'][b]) == int(r['status'])
        assert bool(o['info'][b, 13]) == bool(r['early_exit'])
        assert rel(o['Hc'][b], r['Hc']) < PARITY
        assert abs(o['kappa'][b] - r['kappa']) < 1e-9 * max(1.0, r['kappa'])
        assert rel(o['P'][b], r['P']) < PARITY
    assert np.array_equal(o['status'], s['status'])
    assert np.abs(o['iters'].astype(int) - s['iters'].astype(int)).max() <= 1
    assert rel(o['Hc'], s['Hc']) < 1e-9


def test_persistent_kernel_takes_batches_by_itself():
    """The default rule (TMPC_TUNE_PERSISTENT = 1): a batch that fills the chip goes through the persistent kernel, a single long problem through the launch sequence --
    both equal the forced variants bit for bit; a member that is convex already and an infeasible one inside the batch keep their statuses."""
    from tunempc_amd._lib import HipConvexifier
    nb, p, nx, mb = 192, 20, 3, 2
    A, B, H = co.gen_batch(777, nb, p, nx, mb)
    H[5] = np.eye(nx + mb)[None] * 2.0                      # early exit
    B[7] = 0.0; H[7, :, nx:, nx:] = -np.eye(mb)             # infeasible (convexifier.py:157)
    res = {}
    for mode in (1, 2, 0):
        h = HipConvexifier(p, nx, mb, chunk=nb); h.set_tuning(persistent=mode)
        res[mode] = (h.convexify_batch(A, B, H), h.convexify_batch(A[:1], B[:1], H[:1]))
        h.close()
    for k in ('Hc', 'P', 'kappa', 'status', 'iters'):
        assert np.array_equal(res[1][0][k], res[2][0][k]), k          # 192 problems: persistent by the rule
        assert np.array_equal(res[1][1][k], res[0][1][k]), k          # 1 problem of period 20: launch sequence
    st = res[1][0]['status']
    assert st[5] == 0 and bool(res[1][0]['info'][5, 13]) and st[7] == 2 and (np.delete(st, [5, 7]) == 0).all()
    keep = np.delete(np.arange(nb), [7])                    # (the infeasible member returns wherever its diverging iteration stopped: nothing to compare)
    assert rel(res[2][0]['Hc'][keep], res[0][0]['Hc'][keep]) < 1e-9
    assert res[2][0]['status'][7] == res[0][0]['status'][7] == 2


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(47, 3, 40, 4, 1), (48, 2, 100, 3, 2), (49, 4, 23, 5, 3)])
def test_persistent_kernel_last_update_of_every_stage(seed, nb, p, nx, mb):
    """in k_ipm_small thread 0 ran the control body that may end the problem (PH_DONE) while other waves were still taking their stages through
    update_body, which returns at once on PH_DONE -- a wave handling several stages (p > 16) could skip the LAST update of its later stages.  With TMPC_FLAG_FAST_EXIT
    the skipped update is a whole centering step of the stage (1e-3 ... 1e-2 of Hc); with the barrier both paths stop after the same step of every stage and differ by
    the rounding the not-yet-converged point amplifies (measured 2e-7 ... 5e-6 on these shapes; converged points agree to 1e-10), run to run identically."""
    from tunempc_amd._lib import HipConvexifier, FLAG_FAST_EXIT
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    outs = []
    for mode in (2, 2, 0):
        h = HipConvexifier(p, nx, mb, chunk=nb, flags=FLAG_FAST_EXIT)
        h.set_tuning(persistent=mode)
        outs.append(h.convexify_batch(A, B, H))
        h.close()
    a, a2, s = outs
    assert (a['status'] == 0).all() and np.array_equal(a['status'], s['status']) and np.array_equal(a['iters'], s['iters'])
    assert np.array_equal(a['Hc'], a2['Hc'])                # deterministic
    assert rel(a['Hc'], s['Hc']) < 1e-4 and rel(a['P'], s['P']) < 1e-4
