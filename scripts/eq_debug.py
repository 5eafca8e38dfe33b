import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
nb, p, nx, mb, ng = 512, 64, 24, 8, 2
n = nx + mb
A, B, H = synthetic.gen_batch(100000, 64, p, nx, mb)
A, B, H = (np.concatenate([x] * 8)[:nb] for x in (A, B, H))
rng = np.random.default_rng(1)
G = rng.standard_normal((nb, p, ng, n))
h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng)
out = h.convexify_eq_batch(A, B, H, G)
bad = np.where(~np.isfinite(out['kappa']) | (out['status'] != 0))[0]
print('bad', bad, 'iters hist', np.bincount(out['iters']))
tr = h.trace(nb)
for b in list(bad[:2]) + [int(np.argmax(out['iters']))]:
    print('--- problem', b, 'iters', out['iters'][b], 'status', out['status'][b], 'info', out['info'][b])
    for row in tr[b]:
        if row[0] == 0: break
        print('  it %2d ph %d mu %.3e tau %.8f pinf %.2e dinf %.2e ap %.3f ad %.3f step %.2e shifts %d' % tuple(row))
for mi in (14,):
    h.set_options(max_iter=mi)
    o = h.convexify_eq_batch(A[128:160], B[128:160], H[128:160], G[128:160])
    m = h.debug_multipliers(32, ng)
    b = 16
    print('max_iter', mi, 'iters', o['iters'][b], 'status', o['status'][b], 'phi min %.3e max %.3e z min %.3e max %.3e' % (m['phi'][b].min(), m['phi'][b].max(), m['z'][b].min(), m['z'][b].max()),
          'nonfinite', (~np.isfinite(m['phi'][b])).sum(), (~np.isfinite(m['z'][b])).sum(), 'min ratio dphi/phi %.3e dz/z %.3e' % ((m['dphi'][b] / m['phi'][b]).min(), (m['dz'][b] / m['z'][b]).min()))
    k = np.unravel_index(np.argmin(m['phi'][b]), m['phi'][b].shape); print('   argmin phi', k, m['phi'][b][k], m['z'][b][k], m['dphi'][b][k], m['dz'][b][k])
    k = np.unravel_index(np.argmin(m['z'][b]), m['z'][b].shape); print('   argmin z', k, m['phi'][b][k], m['z'][b][k], m['dphi'][b][k], m['dz'][b][k])

b = 16; d = nx * (nx + 1) // 2; dp = (d + 15) // 16 * 16; nz = ng
psm = h.debug_array(0, b * p * (nz * nz + 6 * nz), p * (nz * nz + 6 * nz)).reshape(p, -1)
pv = h.debug_array(1, b * p * 2 * ng * (2 * n + 2 * nx), p * 2 * ng * (2 * n + 2 * nx)).reshape(p, -1)
dd = h.debug_array(2, b * p * dp, p * dp).reshape(p, dp)
print('psm nonfinite per stage', np.where(~np.isfinite(psm).all(axis=1))[0], 'pvec', np.where(~np.isfinite(pv).all(axis=1))[0], 'Ddiag nonfinite', np.where(~np.isfinite(dd).all(axis=1))[0],
      'Ddiag <= 0 stages', np.where((dd <= 0).any(axis=1))[0])
print('K max', np.abs(psm[:, :4]).max(), 'Ddiag min', dd.min(), 'argmin', np.unravel_index(np.argmin(dd), dd.shape))
Dm = h.debug_array(3, b * p * dp * dp, p * dp * dp).reshape(p, dp, dp)
print('D (factor) nonfinite stages', np.where(~np.isfinite(Dm).all(axis=(1, 2)))[0][:10])
Om = h.debug_array(5, b * p * dp * dp, p * dp * dp).reshape(p, dp, dp)
Fm = h.debug_array(6, b * p * dp * dp, p * dp * dp).reshape(p, dp, dp)
print('O nonfinite stages', np.where(~np.isfinite(Om).all(axis=(1, 2)))[0][:10], 'F nonfinite stages', np.where(~np.isfinite(Fm).all(axis=(1, 2)))[0][:10])
for k in range(p):
    dg = np.diag(Dm[k])
    if not np.isfinite(Dm[k]).all() or dg.max() > 1e8 * max(dd[k].max(), 1):
        big = np.where(dg > 1e8 * np.sqrt(np.abs(dd[k])).max())[0]
        print('stage', k, 'frozen pivots at', big[:10], 'L diag max %.3e' % np.nanmax(dg), 'max |O| %.3e max |F| %.3e' % (np.nanmax(np.abs(Om[k])), np.nanmax(np.abs(Fm[k]))))
print('max |O| per stage (first 30)', ['%.1e' % np.nanmax(np.abs(Om[k])) for k in range(30)])
print('max |F| per stage (20..40)', ['%.1e' % np.nanmax(np.abs(Fm[k])) for k in range(20, 40)])
