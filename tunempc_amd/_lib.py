"""ctypes binding of the C ABI in include/tunempc_hip.h (libtunempc_hip.so, built in-tree by
`__graft_entry__.build()` / `tunempc_amd/build.py`).  No torch types cross this boundary: plain pointers
and sizes.  The loader fails loudly when the library is missing -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INFO_STRIDE = 16
FLAG_NO_MFMA = 1
FLAG_PROFILE = 2
FLAG_FAST_EXIT = 4      # stop after the first full centering step (see include/tunempc_hip.h): faster, not reproducible to 1e-8
STATUS_NAMES = {0: 'Optimal', 1: 'Feasible', 2: 'Infeasible'}

# every symbol declared in include/tunempc_hip.h (the drop-in boundary) ...
ARROW_LD = 32       # tunempc_hip.h: TMPC_ARROW_LD
EXPORTS = [
    'tmpc_device_count', 'tmpc_workspace_bytes', 'tmpc_workspace_bytes_eq', 'tmpc_workspace_bytes_con',
    'tmpc_create', 'tmpc_create_eq', 'tmpc_create_con', 'tmpc_destroy', 'tmpc_get_chunk', 'tmpc_set_options', 'tmpc_set_tight', 'tmpc_set_tuning', 'tmpc_create_ex',
    'tmpc_convexify_batch_host', 'tmpc_convexify_batch_device', 'tmpc_convexify_eq_batch_host', 'tmpc_convexify_step2_batch_host',
    'tmpc_convexify_con_batch_device', 'tmpc_workspace_bytes_step3', 'tmpc_create_step3', 'tmpc_convexify_step3_batch_host', 'tmpc_workspace_bytes_step3_con', 'tmpc_create_step3_con', 'tmpc_convexify_step3_con_batch_host', 'tmpc_convexify_step3_batch_device', 'tmpc_convexify_step3_con_batch_device', 'tmpc_supplement_batch_host', 'tmpc_supplement_terms_batch_host',
    'tmpc_tracking_reference_host', 'tmpc_eig_scan_host', 'tmpc_get_profile', 'tmpc_get_trace', 'tmpc_get_dual_host', 'tmpc_get_dual_con_host', 'tmpc_pack_sensitivities_host', 'tmpc_eig_clip_host',
    'tmpc_last_error', 'tmpc_version',
]
# ... and in include/tunempc_hip_debug.h (unit-test / diagnostic entries)
DEBUG_EXPORTS = [
    'tmpc_debug_gemm_nt', 'tmpc_debug_block_solve', 'tmpc_debug_cr_schedule', 'tmpc_debug_get_multipliers', 'tmpc_debug_get_array',
    'tmpc_debug_min_eig', 'tmpc_debug_min_eig_lane', 'tmpc_debug_factor_bench',
]


def library_path():
    """The in-tree build; TMPC_LIB names another build of the same library (A/B measurements of two builds on one box)."""
    return os.environ.get('TMPC_LIB') or os.path.join(_HERE, 'lib', 'libtunempc_hip.so')


def load_library():
    """Load libtunempc_hip.so (never initialises a device by itself)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"tunempc_amd: HIP library not built ({path}); run `python -c 'import __graft_entry__ as g; g.build()'`"
            " -- there is no CPU fallback for the convexify hot path")
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    lib.tmpc_device_count.restype = C.c_int
    lib.tmpc_workspace_bytes.restype = C.c_uint64
    lib.tmpc_workspace_bytes.argtypes = [C.c_int] * 4
    lib.tmpc_workspace_bytes_eq.restype = C.c_uint64
    lib.tmpc_workspace_bytes_eq.argtypes = [C.c_int] * 5
    lib.tmpc_create_eq.restype = C.c_int
    lib.tmpc_create_eq.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_convexify_eq_batch_host.restype = C.c_int
    lib.tmpc_convexify_eq_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.tmpc_workspace_bytes_con.restype = C.c_uint64
    lib.tmpc_workspace_bytes_con.argtypes = [C.c_int] * 6
    lib.tmpc_create_con.restype = C.c_int
    lib.tmpc_create_con.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_convexify_step2_batch_host.restype = C.c_int
    lib.tmpc_convexify_step2_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp, ip, C.c_double, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.tmpc_workspace_bytes_step3.restype = C.c_uint64
    lib.tmpc_workspace_bytes_step3.argtypes = [C.c_int] * 4
    lib.tmpc_create_step3.restype = C.c_int
    lib.tmpc_create_step3.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_workspace_bytes_step3_con.restype = C.c_uint64
    lib.tmpc_workspace_bytes_step3_con.argtypes = [C.c_int] * 6
    lib.tmpc_create_step3_con.restype = C.c_int
    lib.tmpc_create_step3_con.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_convexify_step3_con_batch_host.restype = C.c_int
    lib.tmpc_convexify_step3_con_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp, ip, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.tmpc_convexify_step3_batch_host.restype = C.c_int
    lib.tmpc_convexify_step3_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.tmpc_debug_get_multipliers.restype = C.c_int
    lib.tmpc_debug_get_multipliers.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp]
    lib.tmpc_debug_get_array.restype = C.c_int
    lib.tmpc_debug_get_array.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, dp]
    lib.tmpc_convexify_con_batch_device.restype = C.c_int
    lib.tmpc_convexify_con_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_double] + [vp] * 10 + [vp]
    lib.tmpc_convexify_step3_batch_device.restype = C.c_int
    lib.tmpc_convexify_step3_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_double] + [vp] * 10 + [vp]
    lib.tmpc_convexify_step3_con_batch_device.restype = C.c_int
    lib.tmpc_convexify_step3_con_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_double] + [vp] * 11 + [vp]
    lib.tmpc_get_dual_con_host.restype = C.c_int
    lib.tmpc_get_dual_con_host.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    lib.tmpc_create.restype = C.c_int
    lib.tmpc_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_destroy.restype = C.c_int
    lib.tmpc_destroy.argtypes = [vp]
    lib.tmpc_get_chunk.restype = C.c_int
    lib.tmpc_get_chunk.argtypes = [vp]
    lib.tmpc_set_tuning.restype = C.c_int
    lib.tmpc_set_tuning.argtypes = [vp, C.c_int, C.c_double]
    lib.tmpc_create_ex.restype = C.c_int
    lib.tmpc_create_ex.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int] * 8
    lib.tmpc_set_tight.restype = C.c_int
    lib.tmpc_set_tight.argtypes = [vp, C.c_int, C.c_double]
    lib.tmpc_set_options.restype = C.c_int
    lib.tmpc_set_options.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.tmpc_convexify_batch_host.restype = C.c_int
    lib.tmpc_convexify_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.tmpc_convexify_batch_device.restype = C.c_int
    lib.tmpc_convexify_batch_device.argtypes = [vp, C.c_int] + [vp] * 12 + [vp]
    lib.tmpc_supplement_batch_host.restype = C.c_int
    lib.tmpc_supplement_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    lib.tmpc_eig_scan_host.restype = C.c_int
    lib.tmpc_eig_scan_host.argtypes = [vp, C.c_int, dp, dp]
    lib.tmpc_eig_clip_host.restype = C.c_int
    lib.tmpc_eig_clip_host.argtypes = [C.c_int, C.c_int, dp, C.c_double, dp, dp, dp, C.POINTER(C.c_int32)]
    lib.tmpc_get_profile.restype = C.c_int
    lib.tmpc_get_profile.argtypes = [vp, dp]
    lib.tmpc_get_trace.restype = C.c_int
    lib.tmpc_get_trace.argtypes = [vp, C.c_int, dp]
    lib.tmpc_pack_sensitivities_host.restype = C.c_int
    lib.tmpc_pack_sensitivities_host.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_int, dp, ip, ip, dp, dp]
    lib.tmpc_get_dual_host.restype = C.c_int
    lib.tmpc_get_dual_host.argtypes = [vp, C.c_int, dp, dp, dp]
    lib.tmpc_debug_gemm_nt.restype = C.c_int
    lib.tmpc_debug_gemm_nt.argtypes = [vp, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tmpc_debug_block_solve.restype = C.c_int
    lib.tmpc_debug_block_solve.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, ip]
    lib.tmpc_supplement_terms_batch_host.restype = C.c_int
    lib.tmpc_supplement_terms_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, dp]
    lib.tmpc_tracking_reference_host.restype = C.c_int
    lib.tmpc_tracking_reference_host.argtypes = [vp, C.c_int, dp, dp, dp, C.c_double, dp, dp, ip]
    lib.tmpc_debug_cr_schedule.restype = C.c_int
    lib.tmpc_debug_cr_schedule.argtypes = [C.c_int, ip, C.c_int]
    lib.tmpc_debug_factor_bench.restype = C.c_int
    lib.tmpc_debug_factor_bench.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    lib.tmpc_debug_min_eig.restype = C.c_int
    lib.tmpc_debug_min_eig.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    lib.tmpc_debug_min_eig_lane.restype = C.c_int
    lib.tmpc_debug_min_eig_lane.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    lib.tmpc_last_error.restype = C.c_char_p
    lib.tmpc_version.restype = C.c_char_p
    _LIB = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


class EigNotConverged(RuntimeError):
    """tmpc_eig_clip_host returned TMPC_E_NOCONV: the Jacobi sweeps were exhausted; `.partial` holds the last iterate."""


E_NOCONV = -6


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"tunempc_amd: {what} failed with code {rc}: {lib.tmpc_last_error().decode()}")


class HipConvexifier:
    """Handle for batched convexification of problems of one shape (p, nx, mb) on the current HIP device."""

    def __init__(self, p, nx, mb, chunk=0, tol=None, center_tol=None, max_iter=None, center_iter=None, flags=0, ng=0, nc=0, step3=False, lanes=0):
        self.lib = load_library()
        if self.lib.tmpc_device_count() < 1:
            raise RuntimeError("tunempc_amd: no HIP device visible; the convexify hot path has no CPU fallback")
        self.p, self.nx, self.mb, self.n = int(p), int(nx), int(mb), int(nx) + int(mb)
        self._h = C.c_void_p()
        self.ng = int(ng)     # rows of the equality-constraint Jacobian per stage (convexifier.py:249-255), 0: none
        self.nc = int(nc)     # room for active-constraint rows per stage (Step 2, convexifier.py:258-266), 0: none
        self.step3 = bool(step3)      # room for the regularisation T_k of Step 3 (convexifier.py:137-147); such a handle also serves the plain model
        if lanes:
            _check(self.lib, self.lib.tmpc_create_ex(C.byref(self._h), int(chunk), self.p, self.nx, self.mb, self.ng, self.nc, int(self.step3), int(lanes)), 'tmpc_create_ex')
        elif self.step3 and (self.ng or self.nc):
            _check(self.lib, self.lib.tmpc_create_step3_con(C.byref(self._h), int(chunk), self.p, self.nx, self.mb, self.ng, self.nc), 'tmpc_create_step3_con')
        elif self.step3:
            _check(self.lib, self.lib.tmpc_create_step3(C.byref(self._h), int(chunk), self.p, self.nx, self.mb), 'tmpc_create_step3')
        else:
            _check(self.lib, self.lib.tmpc_create_con(C.byref(self._h), int(chunk), self.p, self.nx, self.mb, self.ng, self.nc), 'tmpc_create_con')
        self.chunk = int(self.lib.tmpc_get_chunk(self._h))
        self.flags = int(flags)
        self.set_options(tol, center_tol, max_iter, center_iter, flags)

    def set_options(self, tol=None, center_tol=None, max_iter=None, center_iter=None, flags=None):
        if flags is not None:
            self.flags = int(flags)
        _check(self.lib, self.lib.tmpc_set_options(self._h, float(tol or 0.0), float(center_tol or 0.0),
                                                   int(max_iter or 0), int(center_iter or 0), self.flags), 'tmpc_set_options')

    def set_tuning(self, chord_step=None, small_blocks=None, eig_pretest=None, fuse_fwd=None, graph=None, persistent=None, lowp_switch=None):
        """Performance knobs of the handle (include/tunempc_hip.h: tmpc_set_tuning); None keeps the current value."""
        for key, v in ((1, chord_step), (2, small_blocks), (3, eig_pretest), (4, fuse_fwd), (5, graph), (7, persistent), (8, lowp_switch)):
            if v is not None:
                _check(self.lib, self.lib.tmpc_set_tuning(self._h, key, float(v)), 'tmpc_set_tuning')

    def set_tight(self, enable=True, tight_tol=None):
        """Tight-accuracy mode (include/tunempc_hip.h: tmpc_set_tight): continue every Optimal problem towards tight_tol * kappa (default 2^-37)
        with double-double block linear algebra and a dd dual-Newton polish.  Step 1 and Step 2 handles (nx <= 51; with rows of G / C while
        rows * (2 n + 2 nx) <= 4040); Step 3 handles are refused (RuntimeError from TMPC_E_UNSUPPORTED)."""
        _check(self.lib, self.lib.tmpc_set_tight(self._h, 1 if enable else 0, float(tight_tol or 0.0)), 'tmpc_set_tight')

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.tmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ host-buffer entry
    def convexify_batch(self, A, B, H):
        """A [nb,p,nx,nx], B [nb,p,nx,mb], H [nb,p,n,n] (numpy, fp64) -> dict of numpy outputs."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64)
        nb = A.shape[0]
        assert A.shape == (nb, self.p, self.nx, self.nx), A.shape
        assert B.shape == (nb, self.p, self.nx, self.mb), B.shape
        assert H.shape == (nb, self.p, self.n, self.n), H.shape
        out = dict(Hc=np.empty_like(H), dHc=np.empty_like(H), P=np.empty_like(A), alpha=np.empty(nb), beta=np.empty(nb),
                   kappa=np.empty(nb), status=np.empty(nb, np.int32), iters=np.empty(nb, np.int32),
                   info=np.empty((nb, INFO_STRIDE)))
        if nb == 0:
            return out          # empty batch: empty outputs, no device call (the C ABI rejects nb < 1)
        rc = self.lib.tmpc_convexify_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(H), _dptr(out['Hc']), _dptr(out['dHc']),
                                                _dptr(out['P']), _dptr(out['alpha']), _dptr(out['beta']), _dptr(out['kappa']),
                                                _iptr(out['status']), _iptr(out['iters']), _dptr(out['info']))
        _check(self.lib, rc, 'tmpc_convexify_batch_host')
        return out

    def convexify_eq_batch(self, A, B, H, G):
        """Step 1 with the equality-constraint term: G [nb,p,ng,n] (ng of the constructor) -> outputs of convexify_batch + Fg [nb,p,ng]."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64); G = np.ascontiguousarray(G, dtype=np.float64)
        nb = A.shape[0]
        assert A.shape == (nb, self.p, self.nx, self.nx), A.shape
        assert B.shape == (nb, self.p, self.nx, self.mb), B.shape
        assert H.shape == (nb, self.p, self.n, self.n), H.shape
        assert self.ng > 0 and G.shape == (nb, self.p, self.ng, self.n), (G.shape, self.ng)
        out = dict(Hc=np.empty_like(H), dHc=np.empty_like(H), P=np.empty_like(A), Fg=np.empty((nb, self.p, self.ng)),
                   alpha=np.empty(nb), beta=np.empty(nb), kappa=np.empty(nb), status=np.empty(nb, np.int32),
                   iters=np.empty(nb, np.int32), info=np.empty((nb, INFO_STRIDE)))
        if nb == 0:
            return out          # empty batch: empty outputs, no device call (the C ABI rejects nb < 1)
        rc = self.lib.tmpc_convexify_eq_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(H), _dptr(G), _dptr(out['Hc']),
                                                   _dptr(out['dHc']), _dptr(out['P']), _dptr(out['Fg']), _dptr(out['alpha']),
                                                   _dptr(out['beta']), _dptr(out['kappa']), _iptr(out['status']),
                                                   _iptr(out['iters']), _dptr(out['info']))
        _check(self.lib, rc, 'tmpc_convexify_eq_batch_host')
        return out

    def debug_array(self, which, offset, count):
        out = np.empty(int(count))
        _check(self.lib, self.lib.tmpc_debug_get_array(self._h, int(which), int(offset), int(count), _dptr(out)), 'tmpc_debug_get_array')
        return out

    def debug_multipliers(self, nb, nr):
        out = [np.empty((nb, self.p, nr)) for _ in range(4)]
        _check(self.lib, self.lib.tmpc_debug_get_multipliers(self._h, nb, nr, *[_dptr(o) for o in out]), 'tmpc_debug_get_multipliers')
        return dict(phi=out[0], z=out[1], dphi=out[2], dz=out[3])

    def convexify_step2_batch(self, A, B, H, J, ncnt, rho):
        """The Step 2 model (convexifier.py:116-131).  J [nb,p,ng+nc,n]: rows of G_k, then rows of C_k, zero padding;
        ncnt [nb,p] int32: rows of C_k present -> outputs of convexify_batch + FgF [nb,p,ng+nc] (Fg_k, then F_k)."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64); J = np.ascontiguousarray(J, dtype=np.float64)
        ncnt = np.ascontiguousarray(ncnt, dtype=np.int32)
        nb = A.shape[0]
        nr = self.ng + self.nc
        assert A.shape == (nb, self.p, self.nx, self.nx), A.shape
        assert B.shape == (nb, self.p, self.nx, self.mb), B.shape
        assert H.shape == (nb, self.p, self.n, self.n), H.shape
        assert self.nc > 0 and J.shape == (nb, self.p, nr, self.n) and ncnt.shape == (nb, self.p), (J.shape, ncnt.shape, nr)
        out = dict(Hc=np.empty_like(H), dHc=np.empty_like(H), P=np.empty_like(A), FgF=np.empty((nb, self.p, nr)),
                   alpha=np.empty(nb), beta=np.empty(nb), kappa=np.empty(nb), status=np.empty(nb, np.int32),
                   iters=np.empty(nb, np.int32), info=np.empty((nb, INFO_STRIDE)))
        if nb == 0:
            return out          # empty batch: empty outputs, no device call (the C ABI rejects nb < 1)
        rc = self.lib.tmpc_convexify_step2_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(H), _dptr(J), _iptr(ncnt), float(rho),
                                                      _dptr(out['Hc']), _dptr(out['dHc']), _dptr(out['P']), _dptr(out['FgF']),
                                                      _dptr(out['alpha']), _dptr(out['beta']), _dptr(out['kappa']),
                                                      _iptr(out['status']), _iptr(out['iters']), _dptr(out['info']))
        _check(self.lib, rc, 'tmpc_convexify_step2_batch_host')
        return out

    def convexify_step3_batch(self, A, B, H, rho):
        """The Step 3 model (convexifier.py:137-147), plain model + T: outputs of convexify_batch + T [nb,p,n,n] (every entry > 0)."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64)
        nb = A.shape[0]
        assert self.step3, 'handle created without step3=True'
        assert A.shape == (nb, self.p, self.nx, self.nx) and B.shape == (nb, self.p, self.nx, self.mb) and H.shape == (nb, self.p, self.n, self.n)
        out = dict(Hc=np.empty_like(H), dHc=np.empty_like(H), P=np.empty_like(A), T=np.empty_like(H), alpha=np.empty(nb), beta=np.empty(nb),
                   kappa=np.empty(nb), status=np.empty(nb, np.int32), iters=np.empty(nb, np.int32), info=np.empty((nb, INFO_STRIDE)))
        if nb == 0:
            return out
        rc = self.lib.tmpc_convexify_step3_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(H), float(rho), _dptr(out['Hc']), _dptr(out['dHc']),
                                                      _dptr(out['P']), _dptr(out['T']), _dptr(out['alpha']), _dptr(out['beta']), _dptr(out['kappa']),
                                                      _iptr(out['status']), _iptr(out['iters']), _dptr(out['info']))
        _check(self.lib, rc, 'tmpc_convexify_step3_batch_host')
        return out

    def convexify_step3_con_batch(self, A, B, H, J, ncnt, rho):
        """Step 3 with the multipliers of G / C in the same solve (convexifier.py:144): J [nb,p,ng+nc,n] and ncnt [nb,p] as in
        convexify_step2_batch, or J [nb,p,ng,n] with ncnt=None (G only, cost-free multipliers).  Outputs of convexify_batch + FgF + T."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64); J = np.ascontiguousarray(J, dtype=np.float64)
        nb = A.shape[0]
        assert self.step3 and self.ng + self.nc > 0, 'handle created without step3=True and constraint rows'
        nr = self.ng + (self.nc if ncnt is not None else 0)
        assert J.shape == (nb, self.p, nr, self.n), (J.shape, (nb, self.p, nr, self.n))
        if ncnt is not None:
            ncnt = np.ascontiguousarray(ncnt, dtype=np.int32)
            assert ncnt.shape == (nb, self.p)
        out = dict(Hc=np.empty_like(H), dHc=np.empty_like(H), P=np.empty_like(A), T=np.empty_like(H), FgF=np.zeros((nb, self.p, nr)), alpha=np.empty(nb),
                   beta=np.empty(nb), kappa=np.empty(nb), status=np.empty(nb, np.int32), iters=np.empty(nb, np.int32), info=np.empty((nb, INFO_STRIDE)))
        if nb == 0:
            return out
        rc = self.lib.tmpc_convexify_step3_con_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(H), _dptr(J), _iptr(ncnt), float(rho), _dptr(out['Hc']),
                                                          _dptr(out['dHc']), _dptr(out['P']), _dptr(out['FgF']), _dptr(out['T']), _dptr(out['alpha']),
                                                          _dptr(out['beta']), _dptr(out['kappa']), _iptr(out['status']), _iptr(out['iters']), _dptr(out['info']))
        _check(self.lib, rc, 'tmpc_convexify_step3_con_batch_host')
        return out

    # ------------------------------------------------------------------ device-resident entry (torch tensors)
    def convexify_batch_device(self, A, B, H, out=None, stream=None):
        """torch CUDA tensors (fp64, contiguous) in, torch tensors out; data stays in HBM."""
        import torch
        nb = A.shape[0]
        for t in (A, B, H):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        dev = A.device
        if out is None:
            out = dict(Hc=torch.empty_like(H), dHc=torch.empty_like(H), P=torch.empty_like(A),
                       alpha=torch.empty(nb, dtype=torch.float64, device=dev), beta=torch.empty(nb, dtype=torch.float64, device=dev),
                       kappa=torch.empty(nb, dtype=torch.float64, device=dev), status=torch.empty(nb, dtype=torch.int32, device=dev),
                       iters=torch.empty(nb, dtype=torch.int32, device=dev),
                       info=torch.empty((nb, INFO_STRIDE), dtype=torch.float64, device=dev))
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: C.c_void_p(t.data_ptr())
        rc = self.lib.tmpc_convexify_batch_device(self._h, nb, ptr(A), ptr(B), ptr(H), ptr(out['Hc']), ptr(out['dHc']), ptr(out['P']),
                                                  ptr(out['alpha']), ptr(out['beta']), ptr(out['kappa']), ptr(out['status']),
                                                  ptr(out['iters']), ptr(out['info']), C.c_void_p(st))
        _check(self.lib, rc, 'tmpc_convexify_batch_device')
        return out

    def convexify_con_batch_device(self, A, B, H, J, ncnt=None, rho=0.0, stream=None):
        """Device-resident Step 1 with G (ncnt None, J = G [nb,p,ng,n]) or Step 2 model (J [nb,p,ng+nc,n], ncnt [nb,p] int32):
        torch CUDA tensors in, torch tensors out (the dict of convexify_batch_device plus 'FgF')."""
        import torch
        nb = A.shape[0]
        for t in (A, B, H, J):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        assert ncnt is None or (ncnt.is_cuda and ncnt.dtype == torch.int32 and ncnt.is_contiguous())
        nr = self.ng if ncnt is None else self.ng + self.nc
        assert tuple(J.shape) == (nb, self.p, nr, self.n), (tuple(J.shape), nr)
        dev = A.device
        f64 = lambda *sh: torch.empty(sh, dtype=torch.float64, device=dev)
        out = dict(Hc=torch.empty_like(H), dHc=torch.empty_like(H), P=torch.empty_like(A), FgF=f64(nb, self.p, nr), alpha=f64(nb), beta=f64(nb),
                   kappa=f64(nb), status=torch.empty(nb, dtype=torch.int32, device=dev), iters=torch.empty(nb, dtype=torch.int32, device=dev),
                   info=f64(nb, INFO_STRIDE))
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: C.c_void_p(t.data_ptr())
        rc = self.lib.tmpc_convexify_con_batch_device(self._h, nb, ptr(A), ptr(B), ptr(H), ptr(J), ptr(ncnt) if ncnt is not None else None,
                                                      float(rho), ptr(out['Hc']), ptr(out['dHc']), ptr(out['P']), ptr(out['FgF']),
                                                      ptr(out['alpha']), ptr(out['beta']), ptr(out['kappa']), ptr(out['status']),
                                                      ptr(out['iters']), ptr(out['info']), C.c_void_p(st))
        _check(self.lib, rc, 'tmpc_convexify_con_batch_device')
        return out

    def convexify_step3_batch_device(self, A, B, H, rho, J=None, ncnt=None, stream=None):
        """Device-resident Step 3 (convexifier.py:137-147): torch CUDA tensors in, torch tensors out (the dict of convexify_batch_device plus 'T'); with J
        (and ncnt) the multipliers of G / C ride in the same solve (convexifier.py:144), output 'FgF' as well."""
        import torch
        nb = A.shape[0]
        for t in (A, B, H) + ((J,) if J is not None else ()):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        assert self.step3 and (J is None or self.ng + self.nc > 0)
        assert ncnt is None or (ncnt.is_cuda and ncnt.dtype == torch.int32 and ncnt.is_contiguous())
        dev = A.device
        f64 = lambda *sh: torch.empty(sh, dtype=torch.float64, device=dev)
        out = dict(Hc=torch.empty_like(H), dHc=torch.empty_like(H), P=torch.empty_like(A), T=torch.empty_like(H), alpha=f64(nb), beta=f64(nb), kappa=f64(nb),
                   status=torch.empty(nb, dtype=torch.int32, device=dev), iters=torch.empty(nb, dtype=torch.int32, device=dev), info=f64(nb, INFO_STRIDE))
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: C.c_void_p(t.data_ptr())
        if J is None:
            rc = self.lib.tmpc_convexify_step3_batch_device(self._h, nb, ptr(A), ptr(B), ptr(H), float(rho), ptr(out['Hc']), ptr(out['dHc']), ptr(out['P']),
                                                            ptr(out['T']), ptr(out['alpha']), ptr(out['beta']), ptr(out['kappa']), ptr(out['status']),
                                                            ptr(out['iters']), ptr(out['info']), C.c_void_p(st))
            _check(self.lib, rc, 'tmpc_convexify_step3_batch_device')
            return out
        nr = self.ng if ncnt is None else self.ng + self.nc
        assert tuple(J.shape) == (nb, self.p, nr, self.n), (tuple(J.shape), nr)
        out['FgF'] = f64(nb, self.p, nr)
        rc = self.lib.tmpc_convexify_step3_con_batch_device(self._h, nb, ptr(A), ptr(B), ptr(H), ptr(J), ptr(ncnt) if ncnt is not None else None,
                                                            float(rho), ptr(out['Hc']), ptr(out['dHc']), ptr(out['P']), ptr(out['FgF']), ptr(out['T']),
                                                            ptr(out['alpha']), ptr(out['beta']), ptr(out['kappa']), ptr(out['status']), ptr(out['iters']),
                                                            ptr(out['info']), C.c_void_p(st))
        _check(self.lib, rc, 'tmpc_convexify_step3_con_batch_device')
        return out

    def supplement_batch(self, A, B, P):
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        P = np.ascontiguousarray(P, dtype=np.float64)
        nb = A.shape[0]
        dH = np.empty((nb, self.p, self.n, self.n))
        _check(self.lib, self.lib.tmpc_supplement_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(P), _dptr(dH)), 'tmpc_supplement_batch_host')
        return dH

    def supplement_terms_batch(self, A, B, P, J=None, wts=None, T=None):
        """dHc = sym(calH(P) + J' diag(w) J + T) per stage (convexifier.py:165-211); J [nb,p,nr,n], wts [nb,p,nr], T [nb,p,n,n]."""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
        P = np.ascontiguousarray(P, dtype=np.float64)
        nb, nr = A.shape[0], 0
        if J is not None:
            J = np.ascontiguousarray(J, dtype=np.float64); wts = np.ascontiguousarray(wts, dtype=np.float64)
            nr = J.shape[2]
            assert J.shape == (nb, self.p, nr, self.n) and wts.shape == (nb, self.p, nr), (J.shape, wts.shape)
        if T is not None:
            T = np.ascontiguousarray(T, dtype=np.float64)
            assert T.shape == (nb, self.p, self.n, self.n), T.shape
        dH = np.empty((nb, self.p, self.n, self.n))
        _check(self.lib, self.lib.tmpc_supplement_terms_batch_host(self._h, nb, _dptr(A), _dptr(B), _dptr(P), nr, _dptr(J), _dptr(wts), _dptr(T), _dptr(dH)),
               'tmpc_supplement_terms_batch_host')
        return dH

    def eig_scan(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64)
        nb = H.shape[0]
        out = np.empty((nb, self.p, 4))
        _check(self.lib, self.lib.tmpc_eig_scan_host(self._h, nb, _dptr(H), _dptr(out)), 'tmpc_eig_scan_host')
        return out

    def profile(self):
        out = np.zeros(16)
        _check(self.lib, self.lib.tmpc_get_profile(self._h, _dptr(out)), 'tmpc_get_profile')
        keys = ['pre_ms', 'schur_ms', 'factor_ms', 'pass1_ms', 'pass2_ms', 'factor_launches', 'total_ms', 'ipm_iters',
                'problem_factorisations', 'potrf_ms', 'trsm_ms', 'update_ms', 'lanes', 'lowp_factorisations', 'update_f32_ms', 'persistent_problems']
        return dict(zip(keys, out.tolist()))

    def pack_sensitivities(self, C=None, mu=None, Hbig=None, thr=1e-15, ncmax=None, nb=None):
        """GPU form of the array post-processing of Pocp.get_sensitivities (pocp.py:322-361): C [nb,p,nh,n], mu [nb,p,nh] -> C_As
        [nb,p,ncmax,n] (active rows in order, zero-padded), nc [nb,p], idx [nb,p,nh] (-1 padded), q [nb,p,n]; Hbig [nb,p*n,p*n] ->
        H [nb,p,n,n].  Without C: q = zeros (pass nb)."""
        out = {}
        p, n = self.p, self.n
        if C is not None:
            C = np.ascontiguousarray(C, dtype=np.float64); mu = np.ascontiguousarray(mu, dtype=np.float64)
            nb, nh = C.shape[0], C.shape[2]
            assert C.shape == (nb, p, nh, n) and mu.shape == (nb, p, nh), (C.shape, mu.shape)
            ncmax = int(ncmax or nh)
            out.update(C_As=np.empty((nb, p, ncmax, n)), nc=np.empty((nb, p), np.int32), idx=np.empty((nb, p, nh), np.int32))
        else:
            nh, ncmax = 0, 0
            nb = int(nb if nb is not None else Hbig.shape[0])
        out['q'] = np.empty((nb, p, n))
        if Hbig is not None:
            Hbig = np.ascontiguousarray(Hbig, dtype=np.float64)
            assert Hbig.shape == (nb, p * n, p * n), Hbig.shape
            out['H'] = np.empty((nb, p, n, n))
        _check(self.lib, self.lib.tmpc_pack_sensitivities_host(self._h, nb, nh, _dptr(C), _dptr(mu), _dptr(Hbig), float(thr), ncmax,
                                                               _dptr(out.get('C_As')), _iptr(out.get('nc')), _iptr(out.get('idx')),
                                                               _dptr(out['q']), _dptr(out.get('H'))), 'tmpc_pack_sensitivities_host')
        if 'nc' in out and (out['nc'] > ncmax).any():
            raise ValueError('pack_sensitivities: a stage has %d active rows, more than ncmax = %d' % (int(out['nc'].max()), ncmax))
        return out

    def dual(self, nb):
        """Dual iterate of the last wave solved (plain Step 1 model, scaled problem): dict(X1, X2 [nb,p,n,n], x0, tau, alpha, mu_target [nb]);
        see tmpc_get_dual_host -- the data for a solver-independent bound on the optimality gap of kappa."""
        X1 = np.empty((nb, self.p, self.n, self.n)); X2 = np.empty_like(X1); sc = np.empty((nb, 4))
        _check(self.lib, self.lib.tmpc_get_dual_host(self._h, int(nb), _dptr(X1), _dptr(X2), _dptr(sc)), 'tmpc_get_dual_host')
        return dict(X1=X1, X2=X2, x0=sc[:, 0].copy(), tau=sc[:, 1].copy(), alpha=sc[:, 2].copy(), mu_target=sc[:, 3].copy())

    def dual_con(self, nb, arrows=False):
        """Dual side of the stage-local multipliers of the last wave (tmpc_get_dual_con_host): dict(phi, z [nb,p,ng+nc][, aX [nb,p,2,32,32], at [nb,p,2]])."""
        nr = self.ng + self.nc
        phi = np.zeros((nb, self.p, nr)); z = np.zeros_like(phi)
        aX = np.zeros((nb, self.p, 2, ARROW_LD, ARROW_LD)) if arrows else None; at = np.zeros((nb, self.p, 2)) if arrows else None
        _check(self.lib, self.lib.tmpc_get_dual_con_host(self._h, int(nb), _dptr(phi), _dptr(z), _dptr(aX) if arrows else None, _dptr(at) if arrows else None), 'tmpc_get_dual_con_host')
        out = dict(phi=phi, z=z)
        if arrows:
            out.update(aX=aX, at=at)
        return out

    def trace(self, nb):
        """[nb, 80, 10] per-iteration diagnostics of the last chunk (it, phase, mu, tau, pinf, dinf, ap, ad, step, shifts)."""
        out = np.zeros((nb, 80, 10))
        _check(self.lib, self.lib.tmpc_get_trace(self._h, nb, _dptr(out)), 'tmpc_get_trace')
        return out

    # ------------------------------------------------------------------ unit-test hooks
    def debug_gemm_nt(self, Cm, A, B, mode=0, lower=False):
        Cm = np.ascontiguousarray(Cm, dtype=np.float64).copy(); A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        M, N = Cm.shape; K = A.shape[1]
        _check(self.lib, self.lib.tmpc_debug_gemm_nt(self._h, _dptr(Cm), _dptr(A), _dptr(B), M, N, K, int(mode), int(lower)), 'tmpc_debug_gemm_nt')
        return Cm

    def debug_min_eig(self, W, lane=False):
        """smallest eigenvalues of symmetric matrices [nmat, n, n]; lane=True: the one-thread-per-matrix routine of the small shapes (n <= 8)"""
        W = np.ascontiguousarray(W, dtype=np.float64)
        nmat, n, _ = W.shape
        out = np.empty(nmat)
        fn = self.lib.tmpc_debug_min_eig_lane if lane else self.lib.tmpc_debug_min_eig
        _check(self.lib, fn(self._h, nmat, n, _dptr(W), _dptr(out)), 'tmpc_debug_min_eig')
        return out

    def tracking_reference(self, Hc, q, wref, ts):
        """W_k = sym(Hc_k)/ts, yref_k = wref_k - Hc_k^-1 q_k for a stack of stages (pmpc.py:961-974).
        Hc [..., n, n], q/wref [..., n] -> (W [..., n, n], yref [..., n], info [...])."""
        Hc = np.ascontiguousarray(Hc, dtype=np.float64); q = np.ascontiguousarray(q, dtype=np.float64)
        wref = np.ascontiguousarray(wref, dtype=np.float64)
        n = self.nx + self.mb
        if Hc.shape[-2:] != (n, n) or q.shape != Hc.shape[:-1] or wref.shape != q.shape:
            raise ValueError('tracking_reference: expected Hc [..., %d, %d] and q, wref [..., %d]' % (n, n, n))
        ns = int(np.prod(Hc.shape[:-2], dtype=np.int64))
        W = np.empty_like(Hc); yref = np.empty_like(q); info = np.zeros(Hc.shape[:-2], dtype=np.int32)
        _check(self.lib, self.lib.tmpc_tracking_reference_host(self._h, ns, _dptr(Hc), _dptr(q), _dptr(wref), float(ts), _dptr(W), _dptr(yref),
                                                              info.ctypes.data_as(C.POINTER(C.c_int32))), 'tmpc_tracking_reference_host')
        return W, yref, info

    def debug_factor_bench(self, nb, p, d, reps=3):
        """(factorisation ms, single-rhs solve ms) of nb copies of one random SPD block-cyclic-tridiagonal system."""
        out = np.zeros(2)
        _check(self.lib, self.lib.tmpc_debug_factor_bench(self._h, nb, p, d, reps, _dptr(out)), 'tmpc_debug_factor_bench')
        return out

    def debug_block_solve(self, D, Ccpl, rhs):
        D = np.ascontiguousarray(D, dtype=np.float64); Ccpl = np.ascontiguousarray(Ccpl, dtype=np.float64)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        p, d, _ = D.shape
        x = np.empty((p, d)); ns = np.zeros(1, np.int32)
        _check(self.lib, self.lib.tmpc_debug_block_solve(self._h, p, d, _dptr(D), _dptr(Ccpl), _dptr(rhs), _dptr(x), _iptr(ns)), 'tmpc_debug_block_solve')
        return x, int(ns[0])


def eig_clip(A, tol):
    """out = sym(A) + V diag(max(tol - lambda, 0)) V' for one (n x n) or a batch ([nb, n, n]) of symmetric matrices, any n
    (tmpc_eig_clip_host; reference sqp_method.py:327-403).  Returns dict(out, evals, reg, sweeps)."""
    lib = load_library()
    A = np.ascontiguousarray(A, dtype=np.float64)
    single = A.ndim == 2
    A3 = A[None] if single else A
    nb, n, n2 = A3.shape
    if n != n2:
        raise ValueError('square matrices expected')
    out = np.empty_like(A3); ev = np.empty((nb, n)); reg = np.empty(nb); sw = np.zeros(nb, dtype=np.int32)
    rc = lib.tmpc_eig_clip_host(nb, n, _dptr(A3), float(tol), _dptr(out), _dptr(ev), _dptr(reg), _iptr(sw))
    if rc == E_NOCONV:
        err = EigNotConverged(f"tunempc_amd: tmpc_eig_clip_host did not converge: {lib.tmpc_last_error().decode()}")
        err.partial = dict(out=out, evals=ev, reg=reg, sweeps=sw)
        raise err
    _check(lib, rc, 'tmpc_eig_clip_host')
    if single:
        return dict(out=out[0], evals=ev[0], reg=float(reg[0]), sweeps=int(sw[0]))
    return dict(out=out, evals=ev, reg=reg, sweeps=sw)


def cr_schedule(p):
    """The elimination schedule of the block factorisation for period p (host only): dict(prep, levels [(eoff, nelim, uoff, nupd)],
    elim [nelim, 8], upd [nupd, 8], orient [p]) -- see include/tunempc_hip_debug.h."""
    lib = load_library()
    n = lib.tmpc_debug_cr_schedule(int(p), None, 0)
    if n < 0:
        raise ValueError('cr_schedule: p >= 1 expected')
    buf = np.zeros(n, np.int32)
    lib.tmpc_debug_cr_schedule(int(p), _iptr(buf), n)
    nlev, prep, ne, nu = (int(v) for v in buf[:4])
    o = 4
    levels = buf[o:o + 4 * nlev].reshape(nlev, 4); o += 4 * nlev
    elim = buf[o:o + 8 * ne].reshape(ne, 8); o += 8 * ne
    upd = buf[o:o + 8 * nu].reshape(nu, 8); o += 8 * nu
    return dict(prep=prep, levels=levels, elim=elim, upd=upd, orient=buf[o:o + p].copy())
