"""VERDICT r3 item 1, the measurement: how far is the returned Hc from the limit point of the central path, and from what an off-the-shelf
primal-dual solver that stops at a relative gap of 1e-9 returns?  (CPU only: oracle in tight mode, C++ port, dense reference model.)

For each problem: kappa and Hc at the default tolerance 2^-25 and in the tight mode at 2^-33 / 2^-37 / 2^-41 (the last one stands in for the limit
point: fp64 stage arithmetic ends there); the dense restatement of the reference model (oracle/reference_sdp.py, a plain Mehrotra iteration on the
unstructured normal equations that stops as soon as relgap, pinf, dinf < 1e-9 -- what PICOS -> CVXOPT / MOSEK do, no centering at the end).
    python tests/tools/tight_sensitivity.py > profiles/r4_tight_sensitivity.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co  # noqa: E402
import reference_sdp as rs  # noqa: E402
from oracle import cpu_ipm  # noqa: E402

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('distance of Hc to the point at mu_t = 2^-41 kappa (stand-in for the limit of the central path), relative Frobenius norm;')
print('"dense 1e-9": oracle/reference_sdp.py stopped at relgap < 1e-9 (no final centering), the kind of point an off-the-shelf solver returns')
for seed, p, nx, mb in [(10, 4, 3, 2), (11, 8, 4, 1), (5, 16, 4, 1), (3, 12, 6, 3), (100000, 16, 12, 4), (100001, 16, 12, 4)]:
    A, B, H, _, _ = co.gen_problem(seed, p, nx, mb)
    n = nx + mb
    N = 2 * p * n + 1
    pts = {}
    for lt in (25, 33, 37, 41):
        c = cpu_ipm.convexify_batch(A[None], B[None], H[None], tol=2.0 ** -lt, threads=8, tight=(lt > 25))
        pts[lt] = (c['Hc'][0], float(c['kappa'][0]), int(c['status'][0]))
    lim, klim, _ = pts[41]
    line = f'seed {seed:6d} p={p:3d} n={n:2d} N={N:4d}: kappa(2^-41) = {klim:.12f};'
    for lt in (25, 33, 37):
        line += f'  2^-{lt}: Hc {rel(pts[lt][0], lim):.1e} kappa {abs(pts[lt][1] - klim) / klim:.1e} (N tol {N * 2.0 ** -lt:.1e})'
    print(line)
    if p * (nx * (nx + 1) // 2) <= 400:
        Q = [H[k][:nx, :nx] for k in range(p)]; R = [H[k][nx:, nx:] for k in range(p)]; Nm = [H[k][:nx, nx:] for k in range(p)]
        r = rs.solve_step(list(A), list(B), Q, R, Nm, constr=False, tol=1e-9)
        Hd = np.stack(r['Hc'])
        print(f'            dense 1e-9 ({r["solver_status"]}): Hc vs 2^-41 {rel(Hd, lim):.1e}, vs 2^-37 {rel(Hd, pts[37][0]):.1e}, vs default {rel(Hd, pts[25][0]):.1e};  kappa(dense) - kappa(2^-41) = {r["kappa"] - klim:.2e}')
