import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(2, 3, 1)
for (nb, p, d) in [(64, 8, 300), (256, 8, 300), (512, 8, 300), (512, 8, 210), (512, 16, 136)]:
    ms = h.debug_factor_bench(nb, p, d, reps=3)
    d3 = float(d) ** 3
    fl = nb * ((p - 2) * 6.3333 * d3 + 2.3333 * d3 + d3 / 3)
    print(f"nb {nb} p {p} d {d}: v0 {ms[0]:.2f} ms ({fl/ms[0]/1e9:.2f} TF/s)   v1 {ms[1]:.2f} ms ({fl/ms[1]/1e9:.2f} TF/s)", h.lib.tmpc_last_error().decode())
