"""How much could a strictly feasible warm start save?  Upper bound experiment in the CPU oracle: start the IPM from the generator's
hidden feasible point (P_hat, which no real warm start knows) with perfectly centred primal blocks, for several initial barrier
parameters, and count iterations against the cold start."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co
for (p, nx, mb) in [(8, 12, 4), (16, 6, 2), (4, 24, 8)]:
    for seed in range(3):
        A, B, H, Phat, Hhat = co.gen_problem(100000 + seed, p, nx, mb)
        tr = []; r0 = co.sdp_step1(A, B, H, trace=tr)
        n0 = sum(1 for t in tr if t['phase'] == 0)
        s, sbeta = co.auto_scaling(H)
        ev = np.linalg.eigvalsh(Hhat)
        line = f"p={p} n={nx+mb} seed {seed}: cold {r0['iters']} ({n0} main) kappa {r0['kappa']:.4f} | warm from the hidden point:"
        for mu0 in (1e-1, 1e-2, 1e-3, 1e-4):
            alpha = 2.0 / (s * ev.min())                    # M = alpha s Hhat has eigenvalues in [2, 2 cond]
            tau = 2.0 * alpha * s * ev.max()
            tr = []
            try:
                r = co.sdp_step1(A, B, H, opts=dict(warm=dict(P=alpha * s * Phat, alpha=alpha, tau=tau, mu0=mu0)), trace=tr)
                nm = sum(1 for t in tr if t['phase'] == 0)
                line += f"  mu0={mu0:g}: {r['iters']} ({nm} main, dk {abs(r['kappa']-r0['kappa']):.1e})"
            except Exception as e:
                line += f"  mu0={mu0:g}: failed {type(e).__name__}"
        print(line, flush=True)
