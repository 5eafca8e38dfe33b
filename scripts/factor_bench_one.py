import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tunempc_amd._lib import HipConvexifier
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = HipConvexifier(2, 3, 1)
ms = h.debug_factor_bench(nb, 8, 300, reps=1)
print(nb, ms)
