import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(2, 3, 1)
for nb in (64, 512):
    ms = h.debug_factor_bench(nb, 8, 300, reps=3)
    print(f"ABL={os.environ.get('TMPC_ABL','-')} nb {nb}: slot0 {ms[0]:.2f} ms   current {ms[1]:.2f} ms")
