"""Generates the committed golden vectors in tests/golden/*.npz.

The reference stack (PICOS -> CVXOPT/MOSEK) cannot run in this container, so the
vectors hold (a) literal inputs taken from the reference's own example (`examples/convex_lqr.py:40-46`,
data only) and synthetic inputs from the seeded generator of BASELINE.md section 4, and (b) the outputs
of the build's CPU oracle (oracle/convexify_oracle.py) for them, cross-checked by solver-independent
invariants at generation time.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402
import reference_sdp as rs  # noqa: E402

# examples/convex_lqr.py:40-46 (numeric literals of the reference's LQ example)
LQR_A = np.array([[-0.3319, 0.7595, 1.5399], [-0.3393, 0.1250, 0.4245], [-0.5090, 0.9388, 0.8864]])
LQR_B = np.array([[0.1060], [-1.3835], [-0.1496]])
LQR_Q = np.array([[-1.0029, -0.0896, 1.1050], [-0.0896, 1.6790, -0.5762], [1.1050, -0.5762, -0.4381]])
LQR_N = np.array([[-0.0420], [0.2112], [-0.2832]])
LQR_R = np.array([[0.6192]])

CASES = {
    # name: (base_seed, nb, p, nx, mb, kwargs)
    'c2_unicycle_shape': (2000, 2, 30, 4, 1, {}),
    'c3_evaporation_shape': (3000, 3, 50, 2, 2, {}),
    'mid_n16': (4000, 2, 6, 12, 4, {}),
    'awe_shape_n15': (5000, 1, 40, 9, 6, {}),
    'identity_family': (6000, 2, 5, 4, 2, {'identity': True}),
}
# BASELINE.json configs[4]: AWE-shaped synthetic, p=200, n=30 (nx=20, m=10).  One problem: the oracle needs minutes here.
C5_CASES = {
    'c5_awe_synthetic_p200_n30': (8000, 1, 200, 20, 10, {}),
}


def solve_batch(A, B, H):
    out = dict(Hc=[], P=[], kappa=[], alpha=[], beta=[], status=[], iters=[])
    for b in range(A.shape[0]):
        r = co.convexify_arrays(A[b], B[b], H[b])
        inv = co.check_invariants(A[b], B[b], H[b], r)
        assert inv['min_eig'] > 0 and inv['struct_err'] < 1e-12, inv
        if not r['early_exit']:
            assert inv['max_cond'] <= r['kappa'] * (1 + 1e-9), inv
        for k in out:
            out[k].append(r[k])
    return {k: np.array(v) for k, v in out.items()}


def c5():
    for name, (seed, nb, p, nx, mb, kw) in C5_CASES.items():
        A, B, Hs = co.gen_batch(seed, nb, p, nx, mb, **kw)
        o = solve_batch(A, B, Hs)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), A=A, B=B, H=Hs, tol=co.DEFAULT_OPTS['tol'], **o)
        print(name, 'kappa', o['kappa'], 'iters', o['iters'], 'status', o['status'])


def main():
    H = co.build_hessian(LQR_Q, LQR_R, LQR_N)
    A = LQR_A[None, None]; B = LQR_B[None, None]; Hs = H[None, None]
    o = solve_batch(A, B, Hs)
    np.savez(os.path.join(HERE, 'c1_convex_lqr.npz'), A=A, B=B, H=Hs, Q=LQR_Q, R=LQR_R, N=LQR_N, tol=co.DEFAULT_OPTS['tol'], **o)
    print('c1_convex_lqr kappa', o['kappa'], 'iters', o['iters'])
    for name, (seed, nb, p, nx, mb, kw) in CASES.items():
        A, B, Hs = co.gen_batch(seed, nb, p, nx, mb, **kw)
        o = solve_batch(A, B, Hs)
        np.savez(os.path.join(HERE, name + '.npz'), A=A, B=B, H=Hs, tol=co.DEFAULT_OPTS['tol'], **o)
        print(name, 'kappa', o['kappa'], 'iters', o['iters'], 'status', o['status'])
    steps23()


EQ_CASES = {
    # name: (base_seed, nb, p, nx, mb, ng): Step 1 with the equality-constraint term (convexifier.py:249-255, :346-347)
    'eq_term_n5': (7000, 2, 3, 3, 2, 2),
    'eq_term_p1': (20, 2, 1, 3, 1, 1),          # member 1 is already convex
    'eq_term_n9': (7200, 2, 4, 6, 3, 4),
}


def equality_term():
    """Inputs (A, B, H from the seeded generator, G standard normal) and the structured oracle's outputs with Fg."""
    for name, (seed, nb, p, nx, mb, ng) in EQ_CASES.items():
        A, B, Hs = co.gen_batch(seed, nb, p, nx, mb)
        G = np.random.default_rng(seed + 99).standard_normal((nb, p, ng, nx + mb))
        out = dict(Hc=[], P=[], Fg=[], kappa=[], alpha=[], beta=[], status=[], iters=[])
        for b in range(nb):
            r = co.convexify_arrays(A[b], B[b], Hs[b], G=G[b])
            r.setdefault('Fg', np.zeros((p, ng)))          # already-convex member (convexifier.py:83-85): no multipliers
            assert r['status'] == co.STATUS_OPTIMAL and (r['Fg'] >= 0).all()
            ev = np.linalg.eigvalsh(r['Hc'])
            assert ev.min() > 0 and (r['early_exit'] or (ev[:, -1] / ev[:, 0]).max() <= r['kappa'] * (1 + 1e-9))
            assert np.abs(r['Hc'] - Hs[b] - co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b], Fg=r['Fg'])[0]).max() < 1e-12
            for k in out:
                out[k].append(r[k])
        np.savez(os.path.join(HERE, name + '.npz'), A=A, B=B, H=Hs, G=G, tol=co.DEFAULT_OPTS['tol'], **{k: np.array(v) for k, v in out.items()})
        print(name, 'kappa', out['kappa'], 'iters', out['iters'])


STEP2_CASES = {
    # name: (base_seed, nb, p, nx, mb, ng, rows of C_k per stage (0: None), rho): the Step 2 model (convexifier.py:116-131)
    'step2_ragged_n5': (7300, 2, 3, 3, 2, 0, [2, 0, 1], 1e-3),
    'step2_with_g_n6': (7400, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3], 1e-2),
    'step2_p1': (20, 1, 1, 3, 1, 0, [2], 1.0),
}


def step2_inputs(seed, nb, p, nx, mb, ng, ncs):
    n = nx + mb
    A, B, Hs = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 5)
    nc = max(ncs)
    G = rng.standard_normal((nb, p, ng, n))
    C = np.zeros((nb, p, nc, n)); ncnt = np.tile(np.asarray(ncs, np.int32), (nb, 1))
    for b in range(nb):
        for k in range(p):
            C[b, k, :ncs[k]] = rng.standard_normal((ncs[k], n))
    return A, B, Hs, G, C, ncnt


def step2():
    """Inputs and the structured oracle's outputs for the Step 2 model, solved directly (whether or not Step 1 is feasible)."""
    for name, (seed, nb, p, nx, mb, ng, ncs, rho) in STEP2_CASES.items():
        A, B, Hs, G, C, ncnt = step2_inputs(seed, nb, p, nx, mb, ng, ncs)
        out = dict(Hc=[], P=[], F=[], Fg=[], kappa=[], alpha=[], beta=[], status=[], iters=[], objective=[])
        for b in range(nb):
            assert np.linalg.eigvalsh(Hs[b])[:, 0].min() < 0
            Cl = [C[b, k, :ncs[k]] if ncs[k] else None for k in range(p)]
            Gb = G[b] if ng else None
            r = co.sdp_step1(A[b], B[b], Hs[b], G=Gb, C=Cl, rho=rho)
            st, dHc = co.check_convergence(A[b], B[b], Hs[b], r['P'], r['ipm_status'], G=Gb, Fg=r.get('Fg'), C=Cl, F=r['F'])[:2]
            assert st == co.STATUS_OPTIMAL
            Fp = np.zeros((p, max(ncs)))
            for k in range(p):
                if ncs[k]:
                    Fp[k, :ncs[k]] = r['F'][k]
            r.update(Hc=Hs[b] + dHc, F=Fp, status=st, Fg=r.get('Fg', np.zeros((p, 0))))
            for k in out:
                out[k].append(r[k])
        np.savez(os.path.join(HERE, name + '.npz'), A=A, B=B, H=Hs, G=G, C=C, ncnt=ncnt, rho=rho, tol=co.DEFAULT_OPTS['tol'],
                 **{k: np.array(v) for k, v in out.items()})
        print(name, 'kappa', out['kappa'], 'iters', out['iters'], 'objective', out['objective'])


TIGHT_TOL = 2.0 ** -37


def tight():
    """Round 5: vectors of the tight-accuracy mode (sdp_step1(tight=True): continuation to 2^-37 kappa with the block algebra in double-double and the dd dual-Newton
    polish) for the plain model, Step 1 with G and the Step 2 model -- the inputs of 'eq_term_n5' / 'step2_with_g_n6' and a plain pair, with the oracle's outputs in that
    mode; cross-checked at generation time against the dense solver's objective (oracle/reference_sdp.py) and by a second run on inputs 1e-14 apart."""
    def check(A, B, H, r, kw, dense_kw):
        nx = A.shape[1]
        r2 = co.sdp_step1(A * (1 + 1e-14), B, H, dict(tol=TIGHT_TOL, tight=True), **kw)
        assert r['ipm_status'] == 'optimal' and np.abs(r2['P'] - r['P']).max() <= 1e-9 * np.abs(r['P']).max()
        Q = [H[k][:nx, :nx] for k in range(len(A))]; R = [H[k][nx:, nx:] for k in range(len(A))]; N = [H[k][:nx, nx:] for k in range(len(A))]
        d = rs.solve_step(list(A), list(B), Q, R, N, tol=1e-9, **dense_kw)
        obj = r.get('objective', r['beta'])
        assert d['solver_status'] == 'optimal' and abs(obj - d['objective']) <= 5e-8 * obj, (obj, d['objective'])
    # plain
    A, B, Hs = co.gen_batch(3, 2, 6, 4, 2)
    out = dict(Hc=[], P=[], kappa=[], mu_target=[])
    for b in range(2):
        r = co.sdp_step1(A[b], B[b], Hs[b], dict(tol=TIGHT_TOL, tight=True))
        check(A[b], B[b], Hs[b], r, {}, dict(constr=False))
        r['Hc'] = Hs[b] + co.symmetrize(co.calH(A[b], B[b], r['P']))
        for k in out:
            out[k].append(r[k])
    np.savez(os.path.join(HERE, 'tight_plain_n6.npz'), A=A, B=B, H=Hs, tol=TIGHT_TOL, **{k: np.array(v) for k, v in out.items()})
    print('tight_plain_n6 kappa', out['kappa'])
    # Step 1 with G
    seed, nb, p, nx, mb, ng = EQ_CASES['eq_term_n5']
    A, B, Hs = co.gen_batch(seed, nb, p, nx, mb)
    G = np.random.default_rng(seed + 99).standard_normal((nb, p, ng, nx + mb))
    out = dict(Hc=[], P=[], Fg=[], kappa=[], mu_target=[])
    for b in range(nb):
        r = co.sdp_step1(A[b], B[b], Hs[b], dict(tol=TIGHT_TOL, tight=True), G=G[b])
        check(A[b], B[b], Hs[b], r, dict(G=G[b]), dict(G=list(G[b]), constr=False))
        r['Hc'] = Hs[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b], Fg=r['Fg'])[0]
        for k in out:
            out[k].append(r[k])
    np.savez(os.path.join(HERE, 'tight_eq_term_n5.npz'), A=A, B=B, H=Hs, G=G, tol=TIGHT_TOL, **{k: np.array(v) for k, v in out.items()})
    print('tight_eq_term_n5 kappa', out['kappa'])
    # Step 2
    seed, nb, p, nx, mb, ng, ncs, rho = STEP2_CASES['step2_with_g_n6']
    A, B, Hs, G, C, ncnt = step2_inputs(seed, nb, p, nx, mb, ng, ncs)
    out = dict(Hc=[], P=[], F=[], Fg=[], kappa=[], objective=[], mu_target=[])
    for b in range(nb):
        Cl = [C[b, k, :ncs[k]] if ncs[k] else None for k in range(p)]
        r = co.sdp_step1(A[b], B[b], Hs[b], dict(tol=TIGHT_TOL, tight=True), G=G[b], C=Cl, rho=rho)
        check(A[b], B[b], Hs[b], r, dict(G=G[b], C=Cl, rho=rho), dict(G=list(G[b]), C=Cl, rho=rho, constr=True))
        Fp = np.zeros((p, max(ncs)))
        for k in range(p):
            if ncs[k]:
                Fp[k, :ncs[k]] = r['F'][k]
        r.update(Hc=Hs[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], G=G[b], Fg=r['Fg'], C=Cl, F=r['F'])[0], F=Fp)
        for k in out:
            out[k].append(r[k])
    np.savez(os.path.join(HERE, 'tight_step2_with_g_n6.npz'), A=A, B=B, H=Hs, G=G, C=C, ncnt=ncnt, rho=rho, tol=TIGHT_TOL, **{k: np.array(v) for k, v in out.items()})
    print('tight_step2_with_g_n6 kappa', out['kappa'], 'objective', out['objective'])


def tight_step3():
    """The oracle's tight mode on the Step 3 model (second-order cone in the polish): a vector for the kernels to come -- tmpc_set_tight refuses Step 3 handles so far."""
    p, nx, mb, rho = 3, 3, 2, 1e-2
    A, B, Hs = co.gen_batch(501, 2, p, nx, mb)
    out = dict(Hc=[], P=[], T=[], kappa=[], objective=[], mu_target=[])
    for b in range(2):
        r = co.sdp_step1(A[b], B[b], Hs[b], dict(tol=TIGHT_TOL, tight=True), rho=rho, force=True)
        r2 = co.sdp_step1(A[b] * (1 + 1e-14), B[b], Hs[b], dict(tol=TIGHT_TOL, tight=True), rho=rho, force=True)
        assert r['ipm_status'] == 'optimal' and np.abs(r2['T'] - r['T']).max() <= 1e-7 * np.abs(r['T']).max()
        Q = [Hs[b][k][:nx, :nx] for k in range(p)]; R = [Hs[b][k][nx:, nx:] for k in range(p)]; N = [Hs[b][k][:nx, nx:] for k in range(p)]
        d = rs.solve_step(list(A[b]), list(B[b]), Q, R, N, rho=rho, constr=False, force=True, tol=1e-9)
        assert d['solver_status'] == 'optimal' and abs(r['objective'] - d['objective']) <= 5e-8 * r['objective']
        r['Hc'] = Hs[b] + co.convex_hessian_suppl(A[b], B[b], r['P'], T=r['T'])[0]
        for k in out:
            out[k].append(r[k])
    np.savez(os.path.join(HERE, 'tight_step3_n5.npz'), A=A, B=B, H=Hs, rho=rho, tol=TIGHT_TOL, **{k: np.array(v) for k, v in out.items()})
    print('tight_step3_n5 kappa', out['kappa'], 'objective', out['objective'])


def awe_step2():
    """Round 6: the Step 2 model at the REAL AWE shape -- the one constrained example the reference convexifies on a real system (paper p.5:
    p = 40, nx = 9, m = 6, Step 2 with eta_F = 1; examples/awe_system/prepare_inputs.py:86, main.py:56 with Tuner.convexify's default rho = 1.0,
    tuner.py:134): the inputs of 'awe_shape_n15' with ng = 3 rows of G_k and ragged active sets C_k of 0..4 rows (pocp.py:325-339 leaves None where
    no constraint is active), solved by the structured numpy oracle in the default mode and in the tight mode; both cross-checked against
    oracle/cpu_ipm (an independent elimination order) and, for the tight point, by a second run on inputs 1e-14 apart."""
    import cpu_ipm
    seed, nb, p, nx, mb, _ = CASES['awe_shape_n15']
    ng, rho = 3, 1.0
    n = nx + mb
    A, B, Hs = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 5)
    ncs = rng.integers(0, 5, size=p)
    G = rng.standard_normal((nb, p, ng, n)); C = np.zeros((nb, p, 4, n)); ncnt = np.tile(ncs.astype(np.int32), (nb, 1))
    for k in range(p):
        C[0, k, :ncs[k]] = rng.standard_normal((ncs[k], n))
    Cl = [C[0, k, :ncs[k]] if ncs[k] else None for k in range(p)]
    out = {}
    for tag, o in (('', None), ('_tight', dict(tol=TIGHT_TOL, tight=True))):
        r = co.sdp_step1(A[0], B[0], Hs[0], o, G=G[0], C=Cl, rho=rho)
        st, dHc = co.check_convergence(A[0], B[0], Hs[0], r['P'], r['ipm_status'], G=G[0], Fg=r['Fg'], C=Cl, F=r['F'])[:2]
        assert st == co.STATUS_OPTIMAL and r['ipm_status'] == 'optimal'
        Fp = np.zeros((p, 4))
        for k in range(p):
            if ncs[k]:
                Fp[k, :ncs[k]] = r['F'][k]
        if tag:
            r2 = co.sdp_step1(A[0] * (1 + 1e-14), B[0], Hs[0], o, G=G[0], C=Cl, rho=rho)
            assert np.abs(r2['P'] - r['P']).max() <= 1e-9 * np.abs(r['P']).max()
        else:
            c = cpu_ipm.convexify_con_batch(A, B, Hs, np.concatenate([G, C], axis=2), ng=ng, ncnt=ncnt, rho=rho)
            assert np.linalg.norm(c['Hc'][0] - Hs[0] - dHc) <= 2e-9 * np.linalg.norm(Hs[0] + dHc) and abs(c['kappa'][0] - r['kappa']) <= 1e-10 * r['kappa']
        for key, v in (('Hc', Hs[0] + dHc), ('P', r['P']), ('Fg', r['Fg']), ('F', Fp), ('kappa', r['kappa']), ('objective', r['objective']),
                       ('mu_target', r['mu_target']), ('iters', r['iters'])):
            out[key + tag] = np.asarray(v)[None]
        print('awe_step2_n15' + tag, 'kappa', r['kappa'], 'objective', r['objective'], 'iters', r['iters'])
    np.savez(os.path.join(HERE, 'awe_step2_n15.npz'), A=A, B=B, H=Hs, G=G, C=C, ncnt=ncnt, rho=rho, tol=co.DEFAULT_OPTS['tol'], tight_tol=TIGHT_TOL, **out)


def rblock_problem(p=2, nx=2, nu=1, seed=0):
    """B_k = 0, R_k < 0 (Step 1 infeasible: the R block of Hc_k can only come from the constraint / regularisation terms),
    Q_k = I, N_k = 0; Cu = rows reaching exactly the input directions."""
    rng = np.random.default_rng(seed)
    A = [0.5 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) for _ in range(p)]
    B = [np.zeros((nx, nu)) for _ in range(p)]
    Q = [np.eye(nx) for _ in range(p)]; R = [-0.5 * np.eye(nu) for _ in range(p)]; N = [np.zeros((nx, nu)) for _ in range(p)]
    Cu = [np.hstack([np.zeros((nu, nx)), np.eye(nu)]) for _ in range(p)]
    return A, B, Q, R, N, Cu


def steps23():
    """Vectors for the NEXT scope row (Steps 2/3 and the G term, convexifier.py:116-157): inputs + the outputs of the dense
    restatement oracle/reference_sdp.py.  The minimiser is not unique in F/T beyond the optimal value, so what is pinned is
    the step taken, the status, kappa and the objective; dHc is stored for reference."""
    A, B, Q, R, N, Cu = rblock_problem()
    for name, kw, opts in [('n1_step2_active_constraints', dict(C=Cu), {'rho': 1e-3}),
                           ('n1_step1_equality_term', dict(G=Cu), {'rho': 1e-3}),
                           ('n1_step3_force', dict(), {'rho': 1e-3, 'force': True})]:
        dHc, dQc, dRc, dNc, info = rs.convexify_reference(A, B, Q, R, N, opts=opts, **kw)
        res = info['result']
        assert all(np.linalg.eigvalsh(h)[0] > 0 for h in res['Hc'])
        np.savez(os.path.join(HERE, name + '.npz'), A=np.stack(A), B=np.stack(B), Q=np.stack(Q), R=np.stack(R), N=np.stack(N), Cu=np.stack(Cu),
                 rho=opts['rho'], force=bool(opts.get('force', False)), step=info['step'], status=info['status'], kappa=info['kappa'],
                 objective=res['objective'], beta=res['beta'], alpha=res['alpha'], dHc=np.stack(dHc))
        print(name, 'step', info['step'], info['status'], 'kappa', info['kappa'], 'objective', res['objective'])


if __name__ == '__main__':
    if sys.argv[1:] == ['eq']:
        equality_term()          # only the equality-term vectors (the others stay byte-identical)
    elif sys.argv[1:] == ['step2']:
        step2()
    elif sys.argv[1:] == ['c5']:
        c5()
    elif sys.argv[1:] == ['tight']:
        tight()
        tight_step3()
    elif sys.argv[1:] == ['awe']:
        awe_step2()
    else:
        main()
        equality_term()
        step2()
        c5()
        tight()
        tight_step3()
        awe_step2()
