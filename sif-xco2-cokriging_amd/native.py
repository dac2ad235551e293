"""ctypes binding of libcokrige_hip.so (C ABI: include/cokrige.h).

The library is built in-tree by ``build_native.py`` (hipcc, gfx950).  Loading fails
loudly when it is missing -- there is deliberately no numpy fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CK_LIB_PATH: load another build of the library (kernel experiments) instead of overwriting the product one
LIB_PATH = os.environ.get("CK_LIB_PATH") or os.path.join(HERE, "libcokrige_hip.so")

METRIC_HAVERSINE = 0
METRIC_EUCLID = 1
PANEL_SLACK_BYTES = 64 * 512 * 8   # include/cokrige.h: CK_PANEL_SLACK_BYTES
APPLY_SIGMA = 1
APPLY_AUX = 2

_dp = POINTER(c_double)
_lib = None

# name -> argtypes (every function returns int except ck_last_error)
_PROTOS = {
    "ck_version": [],
    "ck_device_count": [POINTER(c_int)],
    "ck_create": [c_int, POINTER(c_void_p)],
    "ck_create_partitioned": [POINTER(c_int), c_int, c_int, POINTER(c_void_p)],
    "ck_destroy": [c_void_p],
    "ck_set_stream": [c_void_p, c_void_p, c_int],
    "ck_set_arena": [c_void_p, c_void_p, c_int64],
    "ck_synchronize": [c_void_p],
    "ck_estimate_bytes": [c_void_p, c_int64, POINTER(c_int64)],
    "ck_set_model": [c_void_p, c_int, _dp, _dp, _dp, _dp, c_double],
    "ck_set_metric": [c_void_p, c_int],
    "ck_set_partition": [c_void_p, c_int, c_int],
    "ck_set_data": [c_void_p, c_int, _dp, _dp, c_int64],
    "ck_distance_dense": [c_void_p, _dp, c_int64, _dp, c_int64, _dp],
    "ck_cov_dense": [c_void_p, c_int, c_int, _dp, c_int64, _dp, c_int64, c_int, _dp],
    "ck_cov_lags": [c_void_p, c_int, c_int, _dp, c_int64, c_int, _dp],
    "ck_model_variogram": [c_void_p, POINTER(c_int32), POINTER(c_int32), _dp, c_int64, c_int, _dp],
    "ck_assemble_joint": [c_void_p],
    "ck_factor": [c_void_p, POINTER(c_int64)],
    "ck_predict": [c_void_p, c_int, _dp, c_int64, _dp, _dp],
    "ck_factor_predict": [c_void_p, c_int, _dp, c_int64, _dp, _dp, POINTER(c_int64)],
    "ck_verify_model": [c_void_p, POINTER(c_int64)],
    "ck_loocv": [c_void_p, c_int, _dp, _dp],
    "ck_sample": [c_void_p, _dp, _dp, c_int64],
    "ck_num_panels": [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int64)],
    "ck_panel_owner": [c_void_p, c_int, POINTER(c_int)],
    "ck_aux_begin": [c_void_p, c_int, _dp, c_int64],
    "ck_panel_factor": [c_void_p, c_int],
    "ck_panel_buffer": [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int64)],
    "ck_panel_apply": [c_void_p, c_int, c_int],
    "ck_panel_apply_sigma": [c_void_p, c_int, c_int, c_int],
    "ck_panel_apply_group": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ck_panel_aux_solve": [c_void_p, c_int],
    "ck_aux_finish": [c_void_p, _dp, _dp],
    "ck_factor_info": [c_void_p, POINTER(c_int64)],
    "ck_predict_local": [c_void_p, c_int, _dp, c_int64, c_double, c_int, _dp, _dp, POINTER(c_int64), POINTER(c_int64),
                         POINTER(c_int64)],
    "ck_local_reserve": [c_void_p, c_int64],
    "ck_vario_begin": [c_void_p, _dp, _dp, c_int64, _dp, _dp, c_int64, c_int],
    "ck_vario_extent": [c_void_p, c_double, _dp, _dp, POINTER(c_int64)],
    "ck_vario_bin": [c_void_p, c_double, _dp, c_int, c_int, _dp, POINTER(c_int64)],
    "ck_vario_end": [c_void_p],
    "ck_vario_stats": [c_void_p, POINTER(c_int64), c_int],
    "ck_ref_distance": [c_int, _dp, _dp, c_int64, _dp],
    "ck_hilbert_order": [_dp, c_int64, POINTER(c_int64)],
    "ck_debug_get_lower": [c_void_p, _dp, c_int64],
    "ck_debug_site_order": [c_void_p, c_int, POINTER(c_int64), c_int64],
    "ck_debug_get_entries": [c_void_p, POINTER(c_int64), POINTER(c_int64), c_int64, _dp],
    "ck_debug_mfma_probe": [c_void_p, POINTER(c_int32)],
    "ck_debug_potrf_profile": [c_void_p, c_int, _dp],
    "ck_debug_coop_profile": [c_void_p, c_int64, _dp],
    "ck_debug_mfma_peak": [c_void_p, c_int, c_int, _dp],
    "ck_debug_gemm_clock": [c_void_p, _dp],
    "ck_debug_stream_overlap": [c_void_p, c_int, c_int64, c_int, _dp],
    "ck_debug_tile_map": [c_int64, c_int, c_int, c_int, POINTER(c_int32), c_int64],
    "ck_debug_tall_map": [c_int64, c_int, c_int, c_int, POINTER(c_int32), c_int64],
    "ck_debug_run_map": [POINTER(c_int32), c_int, POINTER(c_int32), c_int64],
    "ck_debug_gemm_stamps": [c_void_p, POINTER(c_uint64), c_int64, POINTER(c_int64)],
    "ck_debug_cu_probe": [c_void_p, POINTER(c_uint32), c_int, POINTER(c_uint32)],
    "ck_set_option": [c_void_p, c_char_p, c_int64],
    "ck_timings": [c_void_p, _dp, c_int],
    "ck_table_fallbacks": [c_void_p, c_int, POINTER(c_int64)],
    "ck_table_info": [c_void_p, c_int, POINTER(c_int), POINTER(c_int), _dp, _dp, _dp],
    "ck_dev_gemm_nt": [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                       c_int],
}


class NativeError(RuntimeError):
    pass


def exported_names():
    """Every symbol include/cokrige.h declares (used by the symbol-export test)."""
    return ["ck_last_error"] + list(_PROTOS)


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, same as /opt/rocm's) but load it by the unversioned file name,
    so whichever of {torch, this library} comes second would otherwise map a SECOND runtime,
    which then sees no GPU.  When torch is installed, map its copy first (by path, global):
    our NEEDED libamdhip64.so.7 and torch's later dlopen both resolve to it."""
    import importlib.util
    import sys
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


_RET_INT64 = {"ck_debug_tile_map", "ck_debug_tall_map", "ck_debug_run_map"}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python sif-xco2-cokriging_amd/build_native.py` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        L.ck_last_error.restype = c_char_p
        L.ck_last_error.argtypes = []
        for name, args in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = c_int64 if name in _RET_INT64 else c_int
            fn.argtypes = args
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise NativeError(lib().ck_last_error().decode("utf-8", "replace"))


def _f64(a, shape2=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape2 is not None:
        a = a.reshape(-1, shape2)
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def ref_distance(metric: int, A, B):
    """The reference's distance arithmetic for n coordinate pairs on the host (libm), bit for bit
    src/fields.py:332-342 -- what decides variogram ties (include/cokrige.h: ck_ref_distance)."""
    A, B = _f64(A, 2), _f64(B, 2)
    if A.shape != B.shape:
        raise ValueError("A and B must have the same shape")
    out = np.empty(A.shape[0])
    _chk(lib().ck_ref_distance(int(metric), _p(A), _p(B), A.shape[0], _p(out)))
    return out


def tile_map(nvalid: int, J0: int, Jstep: int, nJ: int):
    """(block column, tile row, tile column) of every workgroup of one Cholesky trailing update, in launch order
    (include/cokrige.h: ck_debug_tile_map; host only)."""
    n = lib().ck_debug_tile_map(int(nvalid), int(J0), int(Jstep), int(nJ), None, 0)
    if n < 0:
        _chk(-1)
    out = np.zeros((n, 3), dtype=np.int32)
    if n:
        lib().ck_debug_tile_map(int(nvalid), int(J0), int(Jstep), int(nJ), out.ctypes.data_as(POINTER(c_int32)), n)
    return out


def tall_map(nvalid: int, J0: int, nJ: int, aux_tile_rows: int):
    """(block column, tile row, tile column, is right-hand-side tile) of every workgroup of one update of the tall matrix
    [Sigma; c0^T; z^T] (include/cokrige.h: ck_debug_tall_map; host only)."""
    n = lib().ck_debug_tall_map(int(nvalid), int(J0), int(nJ), int(aux_tile_rows), None, 0)
    if n < 0:
        _chk(-1)
    out = np.zeros((n, 4), dtype=np.int32)
    if n:
        lib().ck_debug_tall_map(int(nvalid), int(J0), int(nJ), int(aux_tile_rows), out.ctypes.data_as(POINTER(c_int32)), n)
    return out


def run_map(counts):
    """(system, unit) of every workgroup of a batched launch over systems with `counts` units each, -1 for padding
    workgroups (include/cokrige.h: ck_debug_run_map; host only)."""
    c = np.ascontiguousarray(counts, dtype=np.int32)
    cp = c.ctypes.data_as(POINTER(c_int32))
    n = lib().ck_debug_run_map(cp, c.size, None, 0)
    if n < 0:
        _chk(-1)
    out = np.zeros((n, 2), dtype=np.int32)
    if n:
        lib().ck_debug_run_map(cp, c.size, out.ctypes.data_as(POINTER(c_int32)), n)
    return out


def hilbert_order(coords):
    """Indices of the sites in the order the library lays them out on the device (include/cokrige.h: ck_hilbert_order;
    host only)."""
    c = _f64(coords, 2)
    perm = np.empty(c.shape[0], dtype=np.int64)
    _chk(lib().ck_hilbert_order(_p(c), c.shape[0], perm.ctypes.data_as(POINTER(c_int64))))
    return perm


def device_count() -> int:
    n = c_int(0)
    _chk(lib().ck_device_count(byref(n)))
    return n.value


class Handle:
    """One GPU, one HIP stream, one cokriging problem (see include/cokrige.h)."""

    def __init__(self, device: int = 0, devices=None, rank: int = 0):
        """device: one GPU, one problem.  devices + rank: this process's handle of a multi-GPU run (one process per
        entry of `devices`), already partitioned (rank, len(devices)) -- include/cokrige.h: ck_create_partitioned."""
        self._h = c_void_p()
        self._keep = []
        if devices is None:
            _chk(lib().ck_create(int(device), byref(self._h)))
        else:
            ids = (c_int * len(devices))(*[int(d) for d in devices])
            _chk(lib().ck_create_partitioned(ids, len(devices), int(rank), byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().ck_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ------------------------------------------------------------------
    def set_stream(self, stream_ptr, external=True):
        """Launch on the caller's HIP stream (0 = the legacy default stream torch uses);
        ``external=False`` restores the handle's own stream."""
        _chk(lib().ck_set_stream(self._h, c_void_p(stream_ptr or 0), int(bool(external))))

    def set_arena(self, dev_ptr: int, nbytes: int, keepalive=None):
        self._keep.append(keepalive)
        _chk(lib().ck_set_arena(self._h, c_void_p(dev_ptr), int(nbytes)))

    def estimate_bytes(self, m: int) -> int:
        out = c_int64(0)
        _chk(lib().ck_estimate_bytes(self._h, int(m), byref(out)))
        return out.value

    def synchronize(self):
        _chk(lib().ck_synchronize(self._h))

    def set_option(self, name: str, value: int):
        _chk(lib().ck_set_option(self._h, name.encode(), int(value)))

    def set_model(self, n_procs, sigma, nu, len_scale, nugget, rho12=0.0):
        s, v, l, g = (_f64(x).ravel() for x in (sigma, nu, len_scale, nugget))
        if n_procs == 2 and (s.size != 2 or v.size != 3 or l.size != 3 or g.size != 2):
            raise ValueError("bivariate model needs sigma[2], nu[3], len_scale[3], nugget[2]")
        _chk(lib().ck_set_model(self._h, int(n_procs), _p(s), _p(v), _p(l), _p(g), float(rho12)))

    def set_metric(self, metric: int):
        _chk(lib().ck_set_metric(self._h, int(metric)))

    def set_partition(self, rank: int, world: int):
        _chk(lib().ck_set_partition(self._h, int(rank), int(world)))

    def set_data(self, k: int, coords, values):
        c = _f64(coords, 2)
        v = _f64(values).ravel()
        if c.shape[0] != v.size:
            raise ValueError("coords and values disagree in length")
        _chk(lib().ck_set_data(self._h, int(k), _p(c), _p(v), c.shape[0]))

    # -- element-wise surface -------------------------------------------------------------
    def distance_dense(self, A, B):
        A, B = _f64(A, 2), _f64(B, 2)
        out = np.empty((A.shape[0], B.shape[0]))
        _chk(lib().ck_distance_dense(self._h, _p(A), A.shape[0], _p(B), B.shape[0], _p(out)))
        return out

    def cov_dense(self, i, j, A, B, use_nugget=True):
        A, B = _f64(A, 2), _f64(B, 2)
        out = np.empty((A.shape[0], B.shape[0]))
        _chk(lib().ck_cov_dense(self._h, int(i), int(j), _p(A), A.shape[0], _p(B), B.shape[0], int(bool(use_nugget)),
                                _p(out)))
        return out

    def cov_lags(self, i, j, h, use_nugget=True):
        h = _f64(h)
        out = np.empty(h.shape)
        _chk(lib().ck_cov_lags(self._h, int(i), int(j), _p(h.ravel()), h.size, int(bool(use_nugget)), _p(out)))
        return out

    def model_variogram(self, i, j, h, kind="semivariogram"):
        """Row-wise model (cross-)variogram values (src/model.py:209-237) in one launch."""
        h = _f64(h).ravel()
        ii = np.ascontiguousarray(np.broadcast_to(np.asarray(i, dtype=np.int32), h.shape))
        jj = np.ascontiguousarray(np.broadcast_to(np.asarray(j, dtype=np.int32), h.shape))
        out = np.empty(h.shape)
        _chk(lib().ck_model_variogram(self._h, ii.ctypes.data_as(POINTER(c_int32)), jj.ctypes.data_as(POINTER(c_int32)),
                                      _p(h), h.size, 1 if kind == "covariogram" else 0, _p(out)))
        return out

    # -- joint path -------------------------------------------------------------------------
    def assemble_joint(self):
        _chk(lib().ck_assemble_joint(self._h))

    def factor(self) -> int:
        info = c_int64(0)
        _chk(lib().ck_factor(self._h, byref(info)))
        return info.value

    def predict(self, i, pcoords):
        pc = _f64(pcoords, 2)
        m = pc.shape[0]
        pred, err = np.empty(m), np.empty(m)
        _chk(lib().ck_predict(self._h, int(i), _p(pc), m, _p(pred), _p(err)))
        return pred, err

    def factor_predict(self, i, pcoords):
        """factor() + predict() with the two sweeps overlapped (include/cokrige.h: ck_factor_predict).  Returns
        (info, pred, err); info != 0: Sigma is not positive definite, pred / err are meaningless."""
        pc = _f64(pcoords, 2)
        m = pc.shape[0]
        pred, err = np.empty(m), np.empty(m)
        info = c_int64(0)
        _chk(lib().ck_factor_predict(self._h, int(i), _p(pc), m, _p(pred), _p(err), byref(info)))
        return info.value, pred, err

    def verify_model(self) -> int:
        """0 if the joint covariance of data and the last predict()'s sites is positive definite, else the index
        of the failing minor (src/joint_prediction.py:260-274; include/cokrige.h: ck_verify_model)."""
        info = c_int64(0)
        _chk(lib().ck_verify_model(self._h, byref(info)))
        return info.value

    def loocv(self, i, n_i):
        pred, err = np.empty(n_i), np.empty(n_i)
        _chk(lib().ck_loocv(self._h, int(i), _p(pred), _p(err)))
        return pred, err

    def sample(self, noise):
        e = _f64(noise).ravel()
        out = np.empty(e.size)
        _chk(lib().ck_sample(self._h, _p(e), _p(out), e.size))
        return out

    # -- step-wise form -----------------------------------------------------------------------
    def num_panels(self):
        n, w, npad = c_int(0), c_int(0), c_int64(0)
        _chk(lib().ck_num_panels(self._h, byref(n), byref(w), byref(npad)))
        return n.value, w.value, npad.value

    def panel_owner(self, K):
        o = c_int(0)
        _chk(lib().ck_panel_owner(self._h, int(K), byref(o)))
        return o.value

    def aux_begin(self, i, pcoords):
        pc = _f64(pcoords, 2)
        self._m = pc.shape[0]
        _chk(lib().ck_aux_begin(self._h, int(i), _p(pc), pc.shape[0]))

    def panel_factor(self, K):
        _chk(lib().ck_panel_factor(self._h, int(K)))

    def panel_buffer(self, K):
        p, nb = c_void_p(), c_int64(0)
        _chk(lib().ck_panel_buffer(self._h, int(K), byref(p), byref(nb)))
        return p.value, nb.value

    def panel_apply(self, K, what):
        _chk(lib().ck_panel_apply(self._h, int(K), int(what)))

    def panel_apply_sigma(self, K, J_lo, J_hi):
        _chk(lib().ck_panel_apply_sigma(self._h, int(K), int(J_lo), int(J_hi)))

    def panel_apply_group(self, K0, np_, what, J_lo, J_hi, phase=0, n_phase=1):
        _chk(lib().ck_panel_apply_group(self._h, int(K0), int(np_), int(what), int(J_lo), int(J_hi), int(phase), int(n_phase)))

    def panel_aux_solve(self, K):
        _chk(lib().ck_panel_aux_solve(self._h, int(K)))

    def aux_finish(self):
        pred, err = np.empty(self._m), np.empty(self._m)
        _chk(lib().ck_aux_finish(self._h, _p(pred), _p(err)))
        return pred, err

    def factor_info(self) -> int:
        info = c_int64(0)
        _chk(lib().ck_factor_info(self._h, byref(info)))
        return info.value

    # -- local-neighbourhood prediction ---------------------------------------------------------------
    def predict_local(self, i, pcoords, max_dist=1e3, cv=False):
        pc = _f64(pcoords, 2)
        m = pc.shape[0]
        pred, err = np.empty(m), np.empty(m)
        ne, npd, km = c_int64(0), c_int64(0), c_int64(0)
        _chk(lib().ck_predict_local(self._h, int(i), _p(pc), m, float(max_dist), int(bool(cv)), _p(pred), _p(err),
                                    byref(ne), byref(npd), byref(km)))
        return pred, err, dict(n_empty=ne.value, n_not_pd=npd.value, k_max=km.value)

    def local_reserve(self, nbytes: int = 0):
        """Pre-size the scratch slab of predict_local (0: the automatic budget) -- include/cokrige.h: ck_local_reserve."""
        _chk(lib().ck_local_reserve(self._h, int(nbytes)))

    # -- empirical variogram ------------------------------------------------------------------------
    def vario_begin(self, coords_i, resid_i, coords_j=None, resid_j=None):
        ci, ri = _f64(coords_i, 2), _f64(resid_i).ravel()
        if coords_j is None:
            _chk(lib().ck_vario_begin(self._h, _p(ci), _p(ri), ci.shape[0], None, None, 0, 1))
        else:
            cj, rj = _f64(coords_j, 2), _f64(resid_j).ravel()
            _chk(lib().ck_vario_begin(self._h, _p(ci), _p(ri), ci.shape[0], _p(cj), _p(rj), cj.shape[0], 0))

    def vario_extent(self, max_dist):
        lo, hi, npos = c_double(0), c_double(0), c_int64(0)
        _chk(lib().ck_vario_extent(self._h, float(max_dist), byref(lo), byref(hi), byref(npos)))
        return lo.value, hi.value, npos.value

    def vario_bin(self, max_dist, edges, covariogram=False):
        e = _f64(edges).ravel()
        nb = e.size - 1
        sums = np.empty(nb)
        counts = np.empty(nb, dtype=np.int64)
        _chk(lib().ck_vario_bin(self._h, float(max_dist), _p(e), e.size, int(bool(covariogram)), _p(sums),
                                counts.ctypes.data_as(POINTER(c_int64))))
        return sums, counts

    def vario_end(self):
        _chk(lib().ck_vario_end(self._h))

    def vario_stats(self):
        out = np.zeros(4, dtype=np.int64)
        _chk(lib().ck_vario_stats(self._h, out.ctypes.data_as(POINTER(c_int64)), 4))
        return dict(zip(["extent_host_pairs", "bin_host_pairs", "bin_visited_pairs", "extent_extra_rounds"], out.tolist()))

    # -- diagnostics -----------------------------------------------------------------------------
    def debug_get_lower(self, n):
        out = np.empty((n, n))
        _chk(lib().ck_debug_get_lower(self._h, _p(out), int(n)))
        return out

    def debug_get_entries(self, rows, cols):
        """Sigma[rows[e], cols[e]] (after assemble_joint) or L[max, min] (after factor); caller's stacked site order."""
        r = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        c = np.ascontiguousarray(cols, dtype=np.int64).ravel()
        if r.size != c.size:
            raise ValueError("rows and cols disagree in length")
        out = np.empty(r.size)
        _chk(lib().ck_debug_get_entries(self._h, r.ctypes.data_as(POINTER(c_int64)), c.ctypes.data_as(POINTER(c_int64)), r.size,
                                        _p(out)))
        return out

    def debug_site_order(self, k, n_k):
        """perm[j] = caller's index (within process k) of the site at internal position j."""
        out = np.empty(int(n_k), dtype=np.int64)
        _chk(lib().ck_debug_site_order(self._h, int(k), out.ctypes.data_as(POINTER(c_int64)), int(n_k)))
        return out

    def potrf_profile(self, iters=200):
        out = np.zeros(8)
        _chk(lib().ck_debug_potrf_profile(self._h, int(iters), _p(out)))
        keys = ["load_us", "factor_us", "scale_store_us", "inv_diag_us", "inv_offdiag_us", "inv_store_us", "launch_prof_us",
                "launch_us"]
        return dict(zip(keys, out.tolist()))

    def coop_profile(self, rows=4096):
        out = np.zeros(64)
        _chk(lib().ck_debug_coop_profile(self._h, int(rows), _p(out)))
        return out

    def mfma_probe(self):
        out = np.empty(64 * 4 * 3, dtype=np.int32)
        _chk(lib().ck_debug_mfma_probe(self._h, out.ctypes.data_as(POINTER(c_int32))))
        return out.reshape(64, 4, 3)

    def table_info(self, block):
        en, ni = c_int(0), c_int(0)
        ql, qh, er = c_double(0), c_double(0), c_double(0)
        _chk(lib().ck_table_info(self._h, int(block), byref(en), byref(ni), byref(ql), byref(qh), byref(er)))
        return dict(enabled=bool(en.value), n_intervals=ni.value, q_lo=ql.value, q_hi=qh.value,
                    max_rel_err=er.value)

    def table_fallbacks(self, reset=True):
        c = c_int64(0)
        _chk(lib().ck_table_fallbacks(self._h, int(bool(reset)), byref(c)))
        return c.value

    def cu_probe(self, mask_words=None, n_wg=4096):
        """(xcc, se, sh, cu) of each workgroup of a launch on a CU-masked stream (None: unmasked)."""
        out = np.zeros(int(n_wg), dtype=np.uint32)
        mp = None
        if mask_words is not None:
            mk = np.ascontiguousarray(mask_words, dtype=np.uint32)
            assert mk.size == 8
            mp = mk.ctypes.data_as(POINTER(c_uint32))
        _chk(lib().ck_debug_cu_probe(self._h, mp, int(n_wg), out.ctypes.data_as(POINTER(c_uint32))))
        return np.stack([out >> 16, (out >> 8) & 7, (out >> 4) & 1, out & 15], axis=1)

    def mfma_peak(self, waves_per_simd=1, iters=20000):
        out = np.zeros(3)
        _chk(lib().ck_debug_mfma_peak(self._h, int(waves_per_simd), int(iters), _p(out)))
        return dict(tflops=out[0], shader_mhz=out[1], cycles_per_mfma_per_wave=out[2])

    def gemm_clock(self):
        """After set_option("gemm_stamps", 1) and a factorisation: the in-kernel clock of the trailing updates."""
        out = np.zeros(6)
        _chk(lib().ck_debug_gemm_clock(self._h, _p(out)))
        return dict(mhz_median=out[0], mhz_p05=out[1], mhz_p95=out[2], workgroups=int(out[3]),
                    wg_cycles_median=out[4], wg_us_median=out[5])

    def stream_overlap(self, mode, rows, n_side):
        out = np.zeros(1 + int(n_side))
        _chk(lib().ck_debug_stream_overlap(self._h, int(mode), int(rows), int(n_side), _p(out)))
        return out

    def gemm_stamps(self, n_workgroups):
        """Raw workgroup stamps of the last stamped trailing update: (n, 4) uint64 and (grid x, grid y, J0, panels)."""
        out = np.zeros((int(n_workgroups), 4), dtype=np.uint64)
        grid = np.zeros(4, dtype=np.int64)
        _chk(lib().ck_debug_gemm_stamps(self._h, out.ctypes.data_as(POINTER(c_uint64)), out.size,
                                        grid.ctypes.data_as(POINTER(c_int64))))
        return out, grid

    def timings(self):
        out = np.zeros(16)
        _chk(lib().ck_timings(self._h, _p(out), 16))
        keys = ["assemble_sigma_ms", "factor_ms", "assemble_aux_ms", "solve_ms", "reduce_ms", "syrk_ms",
                "syrk_launches", "aux_gemm_ms", "aux_gemm_launches", "vario_bin_ms", "local_ms", "verify_ms",
                "panel_coop_redone", "fused_sweeps_ms", "local_alloc_ms", "tall_union_ms"]
        return dict(zip(keys, out.tolist()))

    def dev_gemm_nt(self, C_ptr, ldc, A_ptr, lda, B_ptr, ldb, M, N, K, lower=False):
        _chk(lib().ck_dev_gemm_nt(self._h, c_void_p(C_ptr), ldc, c_void_p(A_ptr), lda, c_void_p(B_ptr), ldb, M, N, K,
                                  int(bool(lower))))
