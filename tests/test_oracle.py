"""CPU tests of the oracle (oracle/convexify_oracle.py): pinned against the reference's example literals,
the dense unstructured cross-check, solver-independent invariants and the committed golden vectors."""
import os

import numpy as np
import pytest
import scipy.linalg as sla

import convexify_oracle as co
import proto_dense


def _lqr():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'c1_convex_lqr.npz'))
    return g


def test_lqr_literals_eigs_and_scaling(golden_dir):
    """eig(H) of examples/convex_lqr.py:40-46 and the autoScaling factors (convexifier.py:374-401)."""
    g = _lqr()
    H = co.build_hessian(g['Q'], g['R'], g['N'])
    ev = np.linalg.eigvalsh(H)
    np.testing.assert_allclose(ev, [-1.89071596, 0.17894836, 0.62538381, 1.94358379], atol=5e-9)
    s, sbeta = co.auto_scaling(H[None])
    assert abs(s - 5.58820422) < 1e-7 and abs(sbeta - 10.86114315) < 1e-7


def test_lqr_feedback_invariance():
    """The reference's only executable assertion (examples/convex_lqr.py:52-58): tuned and indefinite weights
    give the same LQR gain; plus Hc > 0 which the reference checks in check_convergence (:438-451)."""
    g = _lqr()
    A, B, Q, R, N = g['A'][0, 0], g['B'][0, 0], g['Q'], g['R'], g['N']
    dHc, dQc, dRc, dNc = co.convexify(A, B, Q, R, N)
    assert isinstance(dHc, list) and len(dHc) == 1

    def gain(Q_, R_, N_):
        P = sla.solve_discrete_are(A, B, Q_, R_, s=N_)
        return np.linalg.solve(R_ + B.T @ P @ B, B.T @ P @ A + N_.T)

    K = gain(Q, R, N)
    np.testing.assert_allclose(K.ravel(), [0.05129923, 0.23896453, -0.25707625], atol=1e-7)
    Kc = gain(Q + dQc[0], R + dRc[0], N + dNc[0])
    assert np.linalg.norm(K - Kc) < 1e-5
    Hc = co.build_hessian(Q, R, N) + dHc[0]
    assert np.linalg.eigvalsh(Hc).min() > 0


@pytest.mark.parametrize('seed,p,nx,mb', [(0, 3, 3, 2), (20, 1, 3, 1), (30, 2, 3, 1), (3, 5, 4, 2)])
def test_structured_vs_dense(seed, p, nx, mb):
    """Block-cyclic-tridiagonal oracle == unstructured dense IPM on the same SDP (same optimum kappa, same Hc)."""
    A, B, H, _, _ = co.gen_problem(seed, p, nx, mb)
    r = co.convexify_arrays(A, B, H)
    N = 2 * p * (nx + mb) + 1                      # proto_dense's tol is the relative gap = N * (mu/kappa)
    rd = proto_dense.solve(A, B, H, tol=N * co.DEFAULT_OPTS['tol'], verbose=False, center_its=12)
    assert abs(r['kappa'] - rd['tau']) < 5e-5 * r['kappa']
    assert np.linalg.norm(r['Hc'] - rd['Hc']) / np.linalg.norm(r['Hc']) < 1e-4


@pytest.mark.parametrize('seed,p,nx,mb', [(10, 4, 3, 2), (11, 8, 4, 1), (12, 3, 6, 3), (22, 1, 3, 1)])
def test_invariants(seed, p, nx, mb):
    A, B, H, Phat, Hhat = co.gen_problem(seed, p, nx, mb)
    r = co.convexify_arrays(A, B, H)
    inv = co.check_invariants(A, B, H, r)
    assert not r['early_exit']
    assert r['status'] == co.STATUS_OPTIMAL
    assert inv['min_eig'] > 0                                   # Hc positive definite (convexifier.py:442)
    assert inv['struct_err'] < 1e-12                            # Hc - H = sym(calH(P)) (eq. 18 structure)
    assert inv['max_cond'] <= r['kappa'] * (1 + 1e-9)           # cond(Hc_k) <= sbeta*beta
    kap_hat = max(np.linalg.cond(Hhat[k]) for k in range(p))   # a feasible point: optimum cannot be worse
    assert r['kappa'] <= kap_hat * (1 + 1e-6)


def test_known_optimum_identity_family():
    """Hhat = I  =>  kappa* = 1 and Hc = c*I, the one family with a solver-independent answer."""
    A, B, H, Phat, _ = co.gen_problem(77, 5, 4, 2, identity=True)
    r = co.convexify_arrays(A, B, H)
    assert abs(r['kappa'] - 1.0) < 1e-5
    for k in range(5):
        Hk = r['Hc'][k]
        c = np.trace(Hk) / Hk.shape[0]
        assert np.linalg.norm(Hk - c * np.eye(6)) < 2e-5 * c
    np.testing.assert_allclose(r['Hc'], np.broadcast_to(np.eye(6), r['Hc'].shape), atol=5e-5)


def test_early_exit_returns_bare_zero_arrays():
    """convexifier.py:83-85: already-convex input returns four bare zero arrays (not lists)."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((3, 3)); B = rng.standard_normal((3, 1))
    H = np.eye(4) * 2.0
    out = co.convexify(A, B, H[:3, :3], H[3:, 3:], H[:3, 3:])
    assert all(isinstance(o, np.ndarray) for o in out)
    assert out[0].shape == (4, 4) and out[1].shape == (3, 3) and out[2].shape == (1, 1) and out[3].shape == (3, 1)
    assert not any(o.any() for o in out)


def test_infeasible_raises_value_error():
    """B = 0 with R not PD: R-block of Hc can never become PD => ValueError (convexifier.py:157)."""
    A = 0.5 * np.eye(2); B = np.zeros((2, 1))
    Q = np.eye(2); R = np.array([[-1.0]]); N = np.zeros((2, 1))
    with pytest.raises(ValueError, match='Convexification is not possible'):
        co.convexify(A, B, Q, R, N)


def test_input_checks_messages():
    """preprocessing.py:167,171,178."""
    M = np.eye(2)
    with pytest.raises(AssertionError, match='same type'):
        co.input_checks({'A': [M], 'B': M})
    with pytest.raises(AssertionError, match='same length'):
        co.input_checks({'A': [M, M], 'B': [M]})
    with pytest.raises(AssertionError, match='same size'):
        co.input_checks({'A': [M, np.eye(3)], 'B': [M, M]})
    out = co.input_checks({'A': M, 'B': M})
    assert isinstance(out['A'], list) and len(out['A']) == 1


@pytest.mark.parametrize('name', ['c1_convex_lqr', 'c2_unicycle_shape', 'c3_evaporation_shape', 'mid_n16', 'identity_family'])
def test_golden_vectors(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    assert float(g['tol']) == co.DEFAULT_OPTS['tol']
    for b in range(min(g['A'].shape[0], 2)):
        r = co.convexify_arrays(g['A'][b], g['B'][b], g['H'][b])
        assert int(r['status']) == int(g['status'][b])
        assert np.linalg.norm(r['Hc'] - g['Hc'][b]) / np.linalg.norm(g['Hc'][b]) < 1e-8
        assert abs(r['kappa'] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])


@pytest.mark.parametrize('name', ['eq_term_n5', 'eq_term_p1', 'eq_term_n9'])
def test_equality_term_golden_vectors(golden_dir, name):
    """Step 1 with G (convexifier.py:249-255): the oracle reproduces the committed outputs, Fg included."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    b = 0
    r = co.convexify_arrays(g['A'][b], g['B'][b], g['H'][b], G=g['G'][b])
    assert int(r['status']) == int(g['status'][b])
    assert np.linalg.norm(r['Hc'] - g['Hc'][b]) / np.linalg.norm(g['Hc'][b]) < 1e-8
    assert np.linalg.norm(r['Fg'] - g['Fg'][b]) / np.linalg.norm(g['Fg'][b]) < 1e-7
    assert abs(r['kappa'] - g['kappa'][b]) < 1e-9 * max(1.0, g['kappa'][b])


@pytest.mark.parametrize('name', ['step2_ragged_n5', 'step2_with_g_n6', 'step2_p1'])
def test_step2_golden_vectors(golden_dir, name):
    """Step 2 model (convexifier.py:116-131): the oracle reproduces the committed outputs (Hc, F, Fg, objective)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    b = 0
    p = g['A'].shape[1]
    ng = g['G'].shape[2]
    C = [g['C'][b, k, :g['ncnt'][b, k]] if g['ncnt'][b, k] else None for k in range(p)]
    G = g['G'][b] if ng else None
    r = co.sdp_step1(g['A'][b], g['B'][b], g['H'][b], G=G, C=C, rho=float(g['rho']))
    dHc = co.convex_hessian_suppl(g['A'][b], g['B'][b], r['P'], G=G, Fg=r.get('Fg'), C=C, F=r['F'])[0]
    assert np.linalg.norm(g['H'][b] + dHc - g['Hc'][b]) / np.linalg.norm(g['Hc'][b]) < 1e-8
    assert abs(r['kappa'] - g['kappa'][b]) < 1e-9 * g['kappa'][b] and abs(r['objective'] - g['objective'][b]) < 1e-9 * g['objective'][b]
    for k in range(p):
        if C[k] is not None:
            assert np.linalg.norm(r['F'][k] - g['F'][b, k, :len(r['F'][k])]) <= 1e-7 * max(1.0, np.linalg.norm(r['F'][k]))


def test_dropin_oracle_accepts_equality_jacobians():
    """convexify(..., G=...) -> the supplement carries the G' diag(Fg) G term (convexifier.py:196-197)."""
    A, B, H = co.gen_problem(20, 3, 3, 2)[:3]
    G = np.random.default_rng(21).standard_normal((3, 2, 5))
    Q = [h[:3, :3] for h in H]; R = [h[3:, 3:] for h in H]; N = [h[:3, 3:] for h in H]
    dHc, dQc, dRc, dNc = co.convexify(list(A), list(B), Q, R, N, G=list(G))
    r = co.convexify_arrays(A, B, H, G=G)
    assert np.abs(np.stack(dHc) - r['dHc']).max() == 0.0 and np.linalg.eigvalsh(H + np.stack(dHc)).min() > 0


def test_gap_tolerance_reported():
    """kappa - kappa* <= tol*kappa: tightening the tolerance moves kappa by less than the looser gap."""
    A, B, H, _, _ = co.gen_problem(5, 8, 3, 2)
    N = 2 * 8 * 5 + 1
    r5 = co.convexify_arrays(A, B, H, dict(tol=1e-5 / N))
    r4 = co.convexify_arrays(A, B, H, dict(tol=1e-4 / N))
    assert 0 <= r4['kappa'] - r5['kappa'] <= 1.5e-4 * r4['kappa']


def test_supplement_terms_restatement():
    """convex_hessian_suppl with G/C/T terms against the reference's per-stage expressions written out (convexifier.py:190-206)."""
    rng = np.random.default_rng(2)
    p, nx, mb = 3, 3, 2
    n = nx + mb
    A = rng.standard_normal((p, nx, nx)); B = rng.standard_normal((p, nx, mb))
    P = rng.standard_normal((p, nx, nx)); P = P + np.swapaxes(P, 1, 2)
    G = [rng.standard_normal((2, n)) for _ in range(p)]; Fg = [rng.uniform(0, 1, (2, 1)) for _ in range(p)]
    C = [rng.standard_normal((1, n)), None, rng.standard_normal((2, n))]; F = [rng.uniform(0, 1, (1, 1)), None, rng.uniform(0, 1, (2, 1))]
    T = [rng.uniform(0, 1, (n, n)) for _ in range(p)]
    dH, dQ, dR, dN = co.convex_hessian_suppl(A, B, P, G=G, Fg=Fg, C=C, F=F, T=T)
    for i in range(p):
        dP1, dP2 = P[i], P[(i + 1) % p]
        Qco = A[i].T @ dP2 @ A[i] - dP1; Rco = B[i].T @ dP2 @ B[i]; Nco = (B[i].T @ dP2.T @ A[i]).T
        Hco = np.block([[Qco, Nco], [Nco.T, Rco]])
        Hco = Hco + G[i].T @ np.diagflat(Fg[i]) @ G[i]
        if C[i] is not None:
            Hco = Hco + C[i].T @ np.diagflat(F[i]) @ C[i]
        Hco = Hco + T[i]
        Hs = (Hco + Hco.T) / 2
        assert np.allclose(dH[i], Hs, atol=1e-13)
        assert np.array_equal(dQ[i], dH[i][:nx, :nx]) and np.array_equal(dR[i], dH[i][nx:, nx:]) and np.array_equal(dN[i], dH[i][:nx, nx:])


def test_cost_free_multipliers_equal_equality_rows():
    """The beta-only reading of convexifier.py:276-283 (cost_free=True / rho = 0): with the same number of rows at every stage the
    rows of C_k are indistinguishable from rows of G_k; with ragged rows kappa can only be lower than with the norm terms."""
    A, B, H = co.gen_batch(7, 1, 5, 4, 2)
    A, B, H = A[0], B[0], H[0]
    rng = np.random.default_rng(3)
    C = [rng.standard_normal((2, 6)) for _ in range(5)]
    r1 = co.sdp_step1(A, B, H, C=C, cost_free=True)
    r2 = co.sdp_step1(A, B, H, G=np.stack(C))
    assert np.array_equal(r1['P'], r2['P']) and np.array_equal(np.stack(r1['F']), r2['Fg'])
    Cr = [C[0], None, C[2][:1], C[3], None]
    r3 = co.sdp_step1(A, B, H, C=Cr, rho=0.0)
    r4 = co.sdp_step1(A, B, H, C=Cr, rho=1e-3)
    assert r3['ipm_status'] == r4['ipm_status'] == 'optimal' and r3['kappa'] <= r4['kappa'] * (1 + 1e-7)
    assert r3['F'][1] is None and r3['F'][4] is None and all((f >= 0).all() for f in r3['F'] if f is not None)


def test_hard_target_backs_off_and_ends_optimal():
    """cond(Hhat) = 1e5: the Schur complement turns numerically singular at the default mu_t.  The solver aims one power of two
    earlier AND takes the step of the shifted factorisation towards it (round 3) -- repeating the iteration from the same iterate
    (rounds 1-2, opts['backoff_step'] = False) meets the same singular matrix ten times and ends inaccurate at 1024 mu_t."""
    from tunempc_amd import synthetic
    A, B, H = synthetic.gen_problem(7000 + 17 * 2, 30, 4, 1, sigP=10.0, cond_exp=5, rad=0.5)
    r = co.sdp_step1(A, B, H)
    k = np.log2(r['mu_target'] / 2.0 ** np.round(np.log2(2.0 ** -25 * r['kappa'])))
    assert r['ipm_status'] == 'optimal' and 1 <= k <= 5 and r['shift'] > 0.0
    assert co.check_convergence(A, B, H, r['P'], r['ipm_status'])[0] == co.STATUS_OPTIMAL
    old = co.sdp_step1(A, B, H, opts=dict(backoff_step=False))
    assert old['ipm_status'] == 'optimal_inaccurate'
    # the wall met in the MAIN phase (two shifted factorisations in a row before mu reaches 2 mu_t): centre where the iterate stands
    A, B, H = synthetic.gen_problem(7000 + 17 * 5, 30, 4, 1, sigP=10.0, cond_exp=5, rad=0.5)
    r = co.sdp_step1(A, B, H)
    assert r['ipm_status'] == 'optimal' and r['mu_target'] > 2.0 ** -25 * r['kappa'] * 1.5


def test_hard_target_wall_rule_ignores_pinf_after_a_shift():
    """cond(Hhat) = 1e8: two shifted factorisations in a row leave pinf at ~7e-3 when mu arrives at mu_t; the centering phase starts
    anyway (its steps one power of two up remove the residual) instead of ending 'inaccurate' (round 3; the HIP path: k_ctrl_a)."""
    from tunempc_amd import synthetic
    A, B, H = synthetic.gen_problem(7000 + 17 * 1, 8, 16, 4, sigP=10.0, cond_exp=8, rad=0.5)
    r = co.sdp_step1(A, B, H)
    assert r['ipm_status'] == 'optimal' and r['shift'] > 0.0 and r['pinf'] < 1e-6
    assert co.check_convergence(A, B, H, r['P'], r['ipm_status'])[0] == co.STATUS_OPTIMAL


def test_step3_with_cost_free_T_is_degenerate():
    """convexifier.py:283-285 under the reading of `picos.sum(obj, abs(rho*T[i]))` in which the norm term never reaches the solver: T_k is
    then cost-free, and Step 3 degenerates -- any M_k can be produced by T_k alone, so the optimum is kappa* = 1 with M_k = I and the "convexified"
    Hessian is Hc_k = I / (s alpha) for every stage, whatever H was; only the scalar alpha is left, and it is not determined (it drifts with the
    weight as rho -> 0).  Shown on the paper's objective with rho -> 0; this is why the drop-in raises NotImplementedError for force + objective='beta'
    instead of returning c * I."""
    A, B, H, _, _ = co.gen_problem(5, 3, 3, 2)
    cs = []
    for rho in (1e-4, 1e-6):
        r = co.sdp_step1(A, B, H, rho=rho, force=True)
        assert r['ipm_status'] == 'optimal' and r['kappa'] - 1.0 < 1e-5
        Hc = H + co.convex_hessian_suppl(A, B, r['P'], T=r['T'])[0]
        c = np.trace(Hc, axis1=1, axis2=2) / Hc.shape[1]
        assert np.linalg.norm(Hc - c[:, None, None] * np.eye(Hc.shape[1])) / np.linalg.norm(Hc) < 1e-6        # Hc_k = c I: H is gone
        assert np.ptp(c) < 1e-6 * c.mean() and abs(c.mean() * r['s'] * r['alpha'] - 1.0) < 1e-5                # c = 1 / (s alpha), the same for every stage
        cs.append(c.mean())
    assert abs(cs[1] / cs[0] - 1.0) > 0.05                                                                     # ... and alpha is not determined by the SDP
