"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/*.h declares, the
product fails loudly without a device (no CPU fallback), and the reference-interface mirrors behave."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header='tunempc_hip.h'):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tmpc_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from tunempc_amd._lib import load_library, EXPORTS, DEBUG_EXPORTS
    lib = load_library()
    declared = _declared_symbols()
    assert declared, 'no declarations parsed'
    assert sorted(EXPORTS) == declared
    debug = _declared_symbols('tunempc_hip_debug.h')
    assert sorted(DEBUG_EXPORTS) == debug and not [d for d in declared if d.startswith('tmpc_debug')]
    for s in declared + debug:
        assert hasattr(lib, s), s
    assert b'gfx950' in lib.tmpc_version()


def test_workspace_query_and_unsupported_dims():
    from tunempc_amd._lib import load_library
    lib = load_library()
    # c4: 64 stages of d=300 (dp=304): D,O,F alone are 3*64*304^2*8 B = 142 MB per problem, the float32 copies of the O / fill blocks (round 6) 2*64*304*320*4 B = 50 MB
    per = lib.tmpc_workspace_bytes(1, 64, 24, 8)
    assert 192e6 < per < 240e6
    assert lib.tmpc_workspace_bytes(2, 64, 24, 8) > 1.9 * per
    assert lib.tmpc_workspace_bytes(1, 64, 30, 8) > lib.tmpc_workspace_bytes(1, 64, 24, 8)      # n = 38: the generic per-stage kernels (round 4; rounds 1-3: unsupported)
    assert lib.tmpc_workspace_bytes(1, 4, 40, 30) > 0       # n = 70: the plain model runs up to n = 96 since round 5 ...
    assert lib.tmpc_workspace_bytes_con(1, 4, 40, 30, 2, 0) == 0 and lib.tmpc_workspace_bytes_step3(1, 4, 40, 30) == 0      # ... models with rows or Step 3 up to 64
    assert lib.tmpc_workspace_bytes(1, 4, 60, 40) == 0      # n = 100 > 96 unsupported
    assert lib.tmpc_workspace_bytes_step3(1, 2, 40, 24) > 0 and lib.tmpc_workspace_bytes_step3(1, 2, 48, 16) == 0          # Step 3 at n = 64: blocks of 2901 fit the substitution kernels (3168), 3257 do not
    assert lib.tmpc_workspace_bytes_con(1, 4, 30, 10, 2, 0) > 0 and lib.tmpc_workspace_bytes_step3(1, 4, 30, 10) > 0      # (sizes only; creating the Step 3 handle is refused at n > 32)
    assert lib.tmpc_workspace_bytes(1, 0, 4, 1) == 0


def _no_gpu():
    from tunempc_amd._lib import load_library
    return load_library().tmpc_device_count() < 1


@pytest.mark.skipif(not _no_gpu(), reason='a HIP device is visible; the loud-failure path is for GPU-less hosts')
def test_product_fails_loudly_without_device():
    from tunempc_amd._lib import HipConvexifier
    from tunempc_amd import convexifier
    with pytest.raises(RuntimeError, match='no HIP device|no CPU fallback'):
        HipConvexifier(3, 3, 1)
    A = np.eye(2) * 0.5; B = np.ones((2, 1)); Q = -np.eye(2); R = np.eye(1); N = np.zeros((2, 1))
    with pytest.raises(RuntimeError):
        convexifier.convexify(A, B, Q, R, N)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tunempc_amd/ or scripts/ may import or execute it (the checkers that do
    live under tests/tools/; bench.py uses it in the cpu_baseline leg only, __graft_entry__ in smoke() and to check it imports)."""
    for dirpath, _, files in list(os.walk(os.path.join(ROOT, 'tunempc_amd'))) + list(os.walk(os.path.join(ROOT, 'scripts'))):
        for f in files:
            if f.endswith(('.py', '.h', '.hip')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'convexify_oracle' not in txt and 'proto_dense' not in txt and 'tracking_oracle' not in txt and 'reference_sdp' not in txt and 'sqp_oracle' not in txt and 'cpu_ipm' not in txt, (dirpath, f)
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), (dirpath, f)


def test_input_checks_mirror_reference_messages():
    from tunempc_amd import preprocessing
    M = np.eye(2)
    with pytest.raises(AssertionError, match='Input arguments should be of same type!'):
        preprocessing.input_checks({'A': [M], 'B': M})
    with pytest.raises(AssertionError, match='Input data lists should have same length!'):
        preprocessing.input_checks({'A': [M, M], 'B': [M]})
    with pytest.raises(AssertionError, match='Data matrices should have same size along trajectory.'):
        preprocessing.input_checks({'A': [M, np.eye(3)], 'B': [M, M]})
    ragged = preprocessing.input_checks({'A': [M, M], 'C': [np.ones((1, 2)), np.ones((3, 2))]})   # C exempt (:180)
    assert len(ragged['C']) == 2
    wrapped = preprocessing.input_checks({'A': np.matrix(M), 'B': np.matrix(M)})
    assert isinstance(wrapped['A'], list)


def test_mtools():
    from tunempc_amd import mtools
    Q = np.array([[1.0, 2.0], [2.0, 3.0]]); R = np.array([[4.0]]); N = np.array([[5.0], [6.0]])
    H = mtools.buildHessian(Q, R, N)
    np.testing.assert_array_equal(H, [[1, 2, 5], [2, 3, 6], [5, 6, 4]])
    S = np.array([[1.0, 2.0], [4.0, 3.0]])
    np.testing.assert_array_equal(mtools.symmetrize(S), [[1, 3], [3, 3]])


def test_synthetic_generator_matches_oracle_generator():
    import convexify_oracle as co
    from tunempc_amd import synthetic
    a = synthetic.gen_batch(321, 2, 3, 4, 2)
    b = co.gen_batch(321, 2, 3, 4, 2)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_too_many_equality_rows_are_rejected_loudly():
    """The HIP path eliminates at most NG_MAX multipliers per stage; more is an error raised before any device call."""
    from tunempc_amd import convexifier
    A = np.eye(2) * 0.5; B = np.ones((2, 1)); Q = -np.eye(2); R = np.eye(1); N = np.zeros((2, 1))
    with pytest.raises(NotImplementedError):
        convexifier.convexify(A, B, Q, R, N, G=np.ones((convexifier.NG_MAX + 1, 3)))


def test_large_blocks_limits_are_rejected_loudly():
    """nx + nu > 32 runs on the generic per-stage kernels: every model up to 64, the plain model up to 96 (round 5); Step 3 only while its blocks (svec(P) + the
    n(n+1)/2 entries of T_k) fit the LDS image of the substitution kernels (3168 since round 5); everything beyond raises NotImplementedError before any device call."""
    from tunempc_amd import convexifier
    from tunempc_amd._lib import load_library
    lib = load_library()
    assert lib.tmpc_workspace_bytes_step3(1, 2, 30, 10) > 0 and lib.tmpc_workspace_bytes_step3_con(1, 2, 36, 12, 24, 24) > 0      # blocks of 1286 and 1893
    assert lib.tmpc_workspace_bytes_step3(1, 2, 40, 24) > 0 and lib.tmpc_workspace_bytes_step3(1, 2, 48, 16) == 0                 # 2901 fit since round 5, 3257 do not
    assert lib.tmpc_workspace_bytes(1, 2, 63, 1) > 0      # the plain model up to nx = 63 (blocks of 2016)
    assert lib.tmpc_workspace_bytes_con(1, 2, 36, 12, 24, 24) > 0 and lib.tmpc_workspace_bytes_con(1, 2, 36, 12, 32, 0) == 0      # rows: up to 31 + 31
    nx, nu = 48, 16
    Ab = np.tile(np.eye(nx) * 0.5, (1, 2, 1, 1)); Bb = np.ones((1, 2, nx, nu)); Hb = np.tile(np.eye(nx + nu), (1, 2, 1, 1))
    with pytest.raises(NotImplementedError, match='Step 3'):
        convexifier.convexify_step3_batch(Ab, Bb, Hb, 1e-2)
    nx, nu = 50, 20                                       # n = 70: the plain model is accepted (it needs a device from here on), rows and Step 3 are not
    Ab = np.tile(np.eye(nx) * 0.5, (1, 2, 1, 1)); Bb = np.ones((1, 2, nx, nu)); Hb = np.tile(np.eye(nx + nu), (1, 2, 1, 1))
    with pytest.raises(NotImplementedError, match='plain model only'):
        convexifier.convexify_batch(Ab, Bb, Hb, G=np.ones((1, 2, 1, nx + nu)))
    with pytest.raises(NotImplementedError, match='plain model only'):
        convexifier.convexify_step3_batch(Ab, Bb, Hb, 1e-2)
    nx, nu = 60, 40
    A = np.eye(nx) * 0.5; B = np.ones((nx, nu)); Q = -np.eye(nx); R = np.eye(nu); N = np.zeros((nx, nu))
    with pytest.raises(NotImplementedError, match='up to nx \\+ nu = 96'):
        convexifier.convexify(A, B, Q, R, N)


def test_rotate_tuning_mirrors_reference_indexing():
    """pmpc.py:773-781: Href[k][j] = H[(k + j) % Nref]."""
    from tunempc_amd import pmpc
    from oracle import tracking_oracle
    H = [np.full((2, 2), float(i)) for i in range(5)]; q = [np.full((1, 2), 10.0 + i) for i in range(5)]
    Href, qref = pmpc.rotate_tuning(H, q, 7)
    assert len(Href) == 5 and all(len(r) == 7 for r in Href)
    assert Href[3][4][0, 0] == (3 + 4) % 5 and qref[4][6][0, 0] == 10.0 + (4 + 6) % 5
    assert [[m[0, 0] for m in row] for row in Href] == [[m[0, 0] for m in row] for row in tracking_oracle.rotate(H, 7)]
    with pytest.raises(AssertionError):
        pmpc.rotate_tuning(H, q[:-1], 3)


def test_tracking_oracle_closed_form():
    """yref = wref - (H/ts)^-1 q/ts = wref - H^-1 q and W = H/ts (pmpc.py:961-974); ts cancels in yref."""
    from oracle import tracking_oracle
    rng = np.random.default_rng(3)
    X = rng.standard_normal((6, 5, 5)); H = X @ np.swapaxes(X, 1, 2) + 0.5 * np.eye(5)
    q = rng.standard_normal((6, 5)); w = rng.standard_normal((6, 5))
    W1, y1 = tracking_oracle.tracking_reference(H, q, w, 0.1)
    W2, y2 = tracking_oracle.tracking_reference(H, q, w, 7.0)
    assert np.allclose(y1, y2, rtol=0, atol=1e-12) and np.allclose(W1 * 0.1, W2 * 7.0)
    for k in range(6):
        assert np.allclose(H[k] @ (w[k] - y1[k]), q[k], atol=1e-12)
        assert np.allclose(y1[k], w[k] - np.linalg.inv(H[k] / 0.1) @ q[k] / 0.1, atol=1e-12)   # the reference's expression verbatim


def test_sensitivity_postprocessing_mirrors_reference():
    """pocp.py:322-361 restated with explicit loops (the reference's own formulation) vs tunempc_amd.pocp."""
    from tunempc_amd import pocp
    rng = np.random.default_rng(5)
    p, nx, nu, nh = 4, 3, 2, 3
    n = nx + nu
    C = [rng.standard_normal((nh, n)) for _ in range(p)]
    mu = [np.array([0.0, 1e-3, 0.0]), np.zeros(3), np.array([2.0, 0.0, -1e-9]), np.array([1e-16, 0.0, 0.0])]
    C_As, idx = pocp.active_set(C, mu)
    assert idx == [[1], [], [0, 2], []]                          # 1e-16 is below the 1e-15 threshold of pocp.py:73
    assert C_As[1] is None and C_As[3] is None
    assert np.array_equal(C_As[0], C[0][[1]]) and np.array_equal(C_As[2], C[2][[0, 2]])
    q = pocp.cost_gradient(mu, C)
    for k in range(p):
        ref = np.zeros((1, n))
        for i in range(nh):
            ref -= mu[k][i] * C[k][i:i + 1, :]
        assert q[k].shape == (1, n) and np.allclose(q[k], ref, atol=1e-15)
    assert all(np.array_equal(v, np.zeros((1, n))) for v in pocp.cost_gradient(None, None, N=p, n=n))
    Hbig = rng.standard_normal((p * n + 2, p * n + 2))
    Hs = pocp.stage_hessians(Hbig, n, p)
    assert len(Hs) == p and all(np.array_equal(Hs[i], Hbig[i * n:(i + 1) * n, i * n:(i + 1) * n]) for i in range(p))


def test_pack_batch_layout():
    from tunempc_amd import pocp
    rng = np.random.default_rng(6)
    p, nx, nu = 3, 2, 1
    n = nx + nu

    def one(seed, with_c):
        r = np.random.default_rng(seed)
        S = {'A': [r.standard_normal((nx, nx)) for _ in range(p)], 'B': [r.standard_normal((nx, nu)) for _ in range(p)],
             'H': [r.standard_normal((n, n)) for _ in range(p)], 'q': [r.standard_normal((1, n)) for _ in range(p)]}
        if with_c:
            S['C_As'] = [r.standard_normal((2, n)), None, r.standard_normal((1, n))]
        return S
    S0, S1 = one(1, True), one(2, False)
    out = pocp.pack_batch([S0, S1], nx)
    assert out['A'].shape == (2, p, nx, nx) and out['B'].shape == (2, p, nx, nu) and out['H'].shape == (2, p, n, n)
    assert out['A'].flags['C_CONTIGUOUS'] and out['H'].dtype == np.float64
    assert np.array_equal(out['H'][1, 2], S1['H'][2]) and np.array_equal(out['q'][0, 1], S0['q'][1][0])
    assert out['nc'].tolist() == [[2, 0, 1], [0, 0, 0]] and out['C'].shape == (2, p, 2, n)
    assert np.array_equal(out['C'][0, 2, 0], S0['C_As'][2][0]) and not out['C'][0, 2, 1].any() and not out['C'][1].any()
    with pytest.raises(AssertionError, match='same length'):
        pocp.pack_batch([S0, {**S1, 'A': S1['A'][:-1]}], nx)
    with pytest.raises(AssertionError, match='same size along trajectory'):
        pocp.pack_batch([{**S1, 'H': [S1['H'][0], np.eye(n + 1), S1['H'][2]]}], nx)


def test_synthetic_generator_knobs():
    """tunempc_amd.synthetic: the defaults are the benchmark distribution (test above);
    cond_exp / rad are the knobs of scripts/robustness_sweep.py (conditioning of the hidden SPD target, spectral radius of A_k)."""
    from tunempc_amd import synthetic
    A, B, H = synthetic.gen_problem(7, 6, 4, 2, sigP=0.0, cond_exp=5, rad=0.5)          # sigP = 0: H is the hidden target itself
    rho = [np.max(np.abs(np.linalg.eigvals(A[k]))) for k in range(6)]
    assert np.allclose(rho, 0.5)
    ev = np.linalg.eigvalsh(H)
    assert ev.min() > 0 and 1e2 < (ev[:, -1] / ev[:, 0]).max() <= 1e5 * (1 + 1e-9)


def test_header_constants_match_the_python_mirror():
    """the limits and keys that include/tunempc_hip.h publishes are the ones the Python side uses"""
    from tunempc_amd import convexifier, _lib
    src = open(os.path.join(ROOT, 'include', 'tunempc_hip.h')).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'^#define\s+(TMPC_[A-Z0-9_]+)\s+(-?\d+)\b', src, flags=re.M)}
    assert defs['TMPC_MAX_ROWS'] == convexifier.NG_MAX == convexifier.NC_MAX == 31
    assert defs['TMPC_ARROW_LD'] == _lib.ARROW_LD == defs['TMPC_MAX_ROWS'] + 1
    assert [defs[k] for k in ('TMPC_TUNE_CHORD_STEP', 'TMPC_TUNE_SMALL_BLOCKS', 'TMPC_TUNE_EIG_PRETEST', 'TMPC_TUNE_FUSE_FWD', 'TMPC_TUNE_GRAPH',
                             'TMPC_TUNE_PERSISTENT', 'TMPC_TUNE_LOWP_SWITCH')] == [1, 2, 3, 4, 5, 7, 8]      # (the positions HipConvexifier.set_tuning passes)
    assert defs['TMPC_INFO_STRIDE'] == 16
    assert convexifier.N_TUNED == 32 and convexifier.N_ROWS_MAX == 64 and convexifier.N_MAX == 96
    lib = _lib.load_library()
    # the row limit is enforced by the library itself (workspace query: 0 = unsupported), not only by the mirror
    assert lib.tmpc_workspace_bytes_con(1, 3, 4, 2, 31, 31) > 0 and lib.tmpc_workspace_bytes_con(1, 3, 4, 2, 32, 0) == 0 and lib.tmpc_workspace_bytes_con(1, 3, 4, 2, 0, 32) == 0


def test_tight_calls_take_handles_of_their_own(monkeypatch):
    """Host logic of the tight-accuracy mode without a device (a recording stand-in for the handle class): a tight call never takes the shared handle of its shape --
    a plain one without rows, or one with exactly the rows of the call -- the mode is switched off again in a `finally`, and tight together with `force` is refused
    before anything is built (round 5: the mode covers G and Step 2)."""
    from tunempc_amd import convexifier as cv
    made = []

    class Fake:
        def __init__(self, p, nx, mb, chunk=0, ng=0, nc=0, step3=False, **kw):
            self.p, self.nx, self.mb, self.chunk, self.ng, self.nc, self.step3 = p, nx, mb, chunk, ng, nc, step3
            self.calls = []
            made.append(self)

        def set_options(self, **kw): pass
        def set_tuning(self, **kw): self.tuning = kw
        def set_tight(self, on, tol=None): self.calls.append(('tight', bool(on)))
        def close(self): self.calls.append(('close',))

        def _out(self, nb, extra):
            n = self.nx + self.mb
            o = dict(Hc=np.zeros((nb, self.p, n, n)), dHc=np.zeros((nb, self.p, n, n)), P=np.zeros((nb, self.p, self.nx, self.nx)), alpha=np.ones(nb), beta=np.ones(nb),
                     kappa=np.ones(nb), status=np.zeros(nb, np.int32), iters=np.zeros(nb, np.int32), info=np.zeros((nb, 16)))
            o.update(extra)
            return o

        def convexify_batch(self, A, B, H): self.calls.append(('plain',)); return self._out(A.shape[0], {})
        def convexify_eq_batch(self, A, B, H, G): self.calls.append(('eq',)); return self._out(A.shape[0], dict(Fg=np.zeros((A.shape[0], self.p, self.ng))))
        def convexify_step2_batch(self, A, B, H, J, ncnt, rho): self.calls.append(('step2',)); return self._out(A.shape[0], dict(FgF=np.zeros((A.shape[0], self.p, self.ng + self.nc))))

    monkeypatch.setattr(cv, 'HipConvexifier', Fake)
    monkeypatch.setattr(cv, '_HANDLES', type(cv._HANDLES)())
    p, nx, mb, ng, nc = 3, 3, 2, 2, 2
    A = np.zeros((1, p, nx, nx)); B = np.zeros((1, p, nx, mb)); H = np.zeros((1, p, nx + mb, nx + mb)); G = np.zeros((1, p, ng, nx + mb)); C = np.zeros((1, p, nc, nx + mb))
    cnt = np.full((1, p), nc, np.int32)
    cv.convexify_step2_batch(A, B, H, C, cnt, 1e-3, G=G)                    # the shared handle of the shape grows room for C rows
    shared = made[-1]
    cv.convexify_batch(A, B, H, G=G, tight=True)
    h_eq = made[-1]
    assert h_eq is not shared and (h_eq.ng, h_eq.nc) == (ng, 0) and h_eq.calls == [('tight', True), ('eq',), ('tight', False)]
    cv.convexify_batch(A, B, H, tight=True)
    h_plain = made[-1]
    assert h_plain is not shared and h_plain is not h_eq and (h_plain.ng, h_plain.nc) == (0, 0) and h_plain.calls == [('tight', True), ('plain',), ('tight', False)]
    cv.convexify_step2_batch(A, B, H, C, cnt, 1e-3, G=G, tight=True)
    h_s2 = made[-1]
    assert h_s2 is not shared and (h_s2.ng, h_s2.nc) == (ng, nc) and h_s2.calls == [('tight', True), ('step2',), ('tight', False)]
    assert h_s2.tuning == dict(lowp_switch=cv.LOWP_SWITCH_DEFAULT)             # every fetch sets the switch of the single-precision updates: the library default ...
    monkeypatch.setattr(cv, 'LOWP_SWITCH', 0.0)
    cv.convexify_step2_batch(A, B, H, C, cnt, 1e-3, G=G, tight=True)
    assert made[-1] is h_s2 and h_s2.tuning == dict(lowp_switch=0.0)            # ... or the module's (0: fp64 throughout), also on a cached handle
    monkeypatch.setattr(cv, 'LOWP_SWITCH', None)
    n_made = len(made)
    cv.convexify_batch(A, B, H, G=G, tight=True)                              # cached: the same handle again
    assert len(made) == n_made and h_eq.calls[-3:] == [('tight', True), ('eq',), ('tight', False)]
    assert ('tight', True) not in shared.calls                                # the shared handle never saw the mode
    # the step logic of a batch passes the mode to both steps: Step 1 on the handle with the rows of G, Step 2 (members that came back Infeasible) on the one with G and C
    class Infeasible(Fake):
        def convexify_eq_batch(self, A, B, H, G):
            o = Fake.convexify_eq_batch(self, A, B, H, G); o['status'][:] = 2; return o
    monkeypatch.setattr(cv, 'HipConvexifier', Infeasible)
    monkeypatch.setattr(cv, '_HANDLES', type(cv._HANDLES)())
    n0 = len(made)
    o = cv.convexify_steps_batch(A, B, H, G=G, C=C, ncnt=cnt, rho=1e-3, tight=True)
    assert list(o['step']) == [2] and len(made) == n0 + 2
    assert made[n0].calls == [('tight', True), ('eq',), ('tight', False)] and (made[n0].ng, made[n0].nc) == (ng, 0)
    assert made[n0 + 1].calls == [('tight', True), ('step2',), ('tight', False)] and (made[n0 + 1].ng, made[n0 + 1].nc) == (ng, nc)
    monkeypatch.setattr(cv, 'HipConvexifier', Fake)
    n_made = len(made)
    Q = [np.eye(nx)] * p; R = [np.eye(mb)] * p; N = [np.zeros((nx, mb))] * p
    with pytest.raises(NotImplementedError):
        cv.convexify([np.eye(nx)] * p, [np.zeros((nx, mb))] * p, Q, R, N, opts={'tight': True, 'force': True})
    assert len(made) == n_made
