"""The C++/OpenMP CPU baseline (oracle/cpu_ipm, bench.py's cpu_baseline leg) against the numpy oracle: same algorithm, same answers."""
import numpy as np
import pytest

import convexify_oracle as co
import cpu_ipm


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(0, 3, 3, 3, 2), (20, 4, 1, 3, 1), (30, 4, 2, 3, 1), (5, 2, 16, 3, 2), (11, 1, 6, 12, 4)])
def test_cpu_ipm_matches_oracle(seed, nb, p, nx, mb):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    o = cpu_ipm.convexify_batch(A, B, H, threads=2)
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b])
        assert int(o['status'][b]) == int(r['status'])
        assert np.linalg.norm(o['Hc'][b] - r['Hc']) <= 1e-8 * np.linalg.norm(r['Hc'])       # the parity bar of the GPU tests
        assert abs(o['kappa'][b] - r['kappa']) <= 1e-9 * max(1.0, r['kappa'])


def test_cpu_ipm_threads_do_not_change_results():
    A, B, H = co.gen_batch(77, 4, 4, 4, 2)
    o1 = cpu_ipm.convexify_batch(A, B, H, threads=1)
    o4 = cpu_ipm.convexify_batch(A, B, H, threads=4)
    assert np.array_equal(o1['Hc'], o4['Hc']) and np.array_equal(o1['iters'], o4['iters'])


def test_cpu_ipm_infeasible_member():
    A, B, H = co.gen_batch(70, 2, 2, 2, 1)
    A[1] = 0.5 * np.eye(2); B[1] = 0.0
    H[1] = co.build_hessian(np.eye(2), np.array([[-1.0]]), np.zeros((2, 1)))
    o = cpu_ipm.convexify_batch(A, B, H)
    assert int(o['status'][1]) == 2 and int(o['status'][0]) == 0


def test_cpu_ipm_hard_targets_back_off_like_the_oracle():
    """cond(Hhat) = 1e5 (scripts/robustness_sweep.py): the Schur matrix turns numerically singular before the default mu_t.  Both
    restatements back mu_t off by powers of two, taking the step of the shifted factorisation, and end Optimal (round 3)."""
    from tunempc_amd import synthetic
    for b in (2, 5):
        A, B, H = synthetic.gen_problem(7000 + 17 * b, 30, 4, 1, sigP=10.0, cond_exp=5, rad=0.5)
        o = cpu_ipm.convexify_batch(A[None], B[None], H[None])
        r = co.convexify_arrays(A, B, H)
        assert int(o['status'][0]) == int(r['status']) == 0
        assert abs(o['kappa'][0] - r['kappa']) <= 1e-4 * r['kappa']              # within N mu_t of each other whatever the back-off count
