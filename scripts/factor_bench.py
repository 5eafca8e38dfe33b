"""Isolated timing of the cyclic-reduction block factorisation and of one solve (tmpc_debug_factor_bench)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier          # (TMPC_LIB=path: an experimental build of the library)
h = HipConvexifier(2, 3, 1)
if os.environ.get('FB_LOWP'):
    h.set_tuning(lowp_switch=2.0)      # (a switch >= 1 in the debug entries: every Schur-complement update in single precision)
if float(os.environ.get('FB_LOWP_VALUE', '0')) > 0:
    h.set_tuning(lowp_switch=float(os.environ['FB_LOWP_VALUE']))      # (scripts/gpu_r6_ablate.sh)
cases = [(512, 8, 300), (64, 8, 300), (8, 8, 300), (512, 64, 300), (64, 64, 300), (8, 64, 300), (1, 64, 300), (64, 200, 210), (1, 30, 10)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (nb, p, d) in cases:
    ms = h.debug_factor_bench(nb, p, d, reps=3)
    d3 = float(d) ** 3
    fl = nb * (max(p - 2, 0) * 6.3333 * d3 + 2.3333 * d3 + d3 / 3)
    byt = nb * p * 5 * 0.5 * (2 * d * d * 8)        # one solve streams D (lower), and the two O blocks of every node twice
    print(f"nb {nb:4d} p {p:3d} d {d:3d}: factor {ms[0]:8.3f} ms ({fl/ms[0]/1e9:6.2f} TF/s algorithmic)   solve {ms[1]:7.3f} ms ({byt/ms[1]/1e9:6.2f} TB/s)",
          h.lib.tmpc_last_error().decode(), flush=True)
