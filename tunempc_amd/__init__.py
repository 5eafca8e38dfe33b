"""tunempc_amd -- MI355X (gfx950) native implementation of TuneMPC's convexify() hot path.

Drop-in surface (same names / argument meaning / errors as the reference):
    tunempc_amd.convexifier.convexify(A, B, Q, R, N, G=None, C=None, opts=...)   (convexifier.py:36-163)
    tunempc_amd.tuner.tuner_convexify(S, nx, p, rho, force, solver)              (tuner.py:134-160)
    tunempc_amd.pocp.active_set / stage_hessians / cost_gradient / pack_batch                   (pocp.py:322-361)
    tunempc_amd.pmpc.tracking_reference(Hc, q, wref, ts), rotate_tuning(H, q, N)      (pmpc.py:773-781,961-974)
plus the batched entry  tunempc_amd.convexifier.convexify_batch(A, B, H, ...).
All arithmetic runs in hand-written HIP kernels behind a C ABI (include/tunempc_hip.h); there is no CPU
fallback: if the shared library or a gfx950 device is missing the calls raise.
"""
from . import mtools, preprocessing, convexifier, pmpc, pocp  # noqa: F401
from ._lib import load_library, library_path, HipConvexifier  # noqa: F401

__version__ = "0.1.0"
