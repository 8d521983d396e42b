"""ctypes binding of libcosmo_hip.so (include/cosmo_hip.h).

This is the same C ABI a Julia `ccall` layer binds (julia/CosmoHIP.jl); the Python binding
exists because Julia is not installed in this image.  There is NO CPU fallback: if the shared library is
missing, or no HIP device is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcosmo_hip.so")            # cosmo_hip_real = double (COSMO.Model{Float64})
LIB_PATH_F32 = os.path.join(_HERE, "libcosmo_hip_f32.so")     # the same sources, cosmo_hip_real = float (COSMO.Model{Float32})

OK = 0
ERR_NAMES = {1: "INVALID", 2: "HIP", 5: "EIG", 6: "UNSUPPORTED", 7: "COMM"}

ZERO, NONNEG, BOX, SOC, PSD_SQUARE, PSD_TRIANGLE = 0, 1, 2, 3, 4, 5
EXP, DUAL_EXP, POW, DUAL_POW = 6, 7, 8, 9
PSD_TRIANGLE_COMPLEX = 10
CUSTOM = 11
KKT_CG, KKT_MINRES_REDUCED, KKT_MINRES, KKT_CG_SR, KKT_CG_JACOBI = 0, 1, 2, 3, 4
STATUS_NAMES = {0: "Undetermined", 1: "Solved", 2: "Max_iter_reached", 3: "Unsolved", 4: "Primal_infeasible",
                5: "Dual_infeasible", 6: "Time_limit_reached"}
MAT_A, MAT_AT, MAT_P, MAT_OP = 0, 1, 2, 3
MAX_RHO_UPDATES = 64
NUM_KERNEL_CLASSES = 16


from ._abi_structs import ABI_VERSION, AccelParams, Params, ResultStruct   # noqa: E402  generated from include/cosmo_hip.h (tools/gen_abi_structs.py)

ACCEL_EMPTY, ACCEL_ANDERSON = 0, 1
PSD_PROJECTION_SIGN, PSD_PROJECTION_EIGEN = 0, 1
# the non-default variants of docs/src/acceleration.md:23 (include/cosmo_hip.h: COSMO_HIP_ACCEL_ANDERSON_*)
ACCEL_ANDERSON_TYPE1_RESTARTED, ACCEL_ANDERSON_TYPE1_ROLLING, ACCEL_ANDERSON_TYPE2NE_RESTARTED, ACCEL_ANDERSON_TYPE2NE_ROLLING = 2, 3, 4, 5


class CosmoHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcosmo_hip error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


_libs = {}

_PD = C.POINTER(C.c_double)
_PF = C.POINTER(C.c_float)


class _RealPtr:
    """Placeholder in SIGNATURES for `cosmo_hip_real*` (include/cosmo_hip.h): resolved to double* or float* when a library is loaded."""


_PR = _RealPtr
_PI64 = C.POINTER(C.c_int64)
_PI32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes).  Kept in one table so tests can check it against the header.
PROJECT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)                       # cosmo_hip_project_fn
CONE_TEST_FN = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_double), C.c_int64, C.c_double, C.c_void_p)     # cosmo_hip_cone_test_fn
PROJECT_FN_F32 = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int64, C.c_void_p)
CONE_TEST_FN_F32 = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_float), C.c_int64, C.c_double, C.c_void_p)

SIGNATURES = {
    "cosmo_hip_version": (C.c_int32, []),
    "cosmo_hip_default_params": (None, [C.POINTER(Params)]),
    "cosmo_hip_create": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32]),
    "cosmo_hip_destroy": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "cosmo_hip_set_problem": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, _PI64, _PI64, _PR, _PI64, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_set_cones": (C.c_int32, [C.c_void_p, C.c_int64, _PI32, _PI64, _PR, _PR]),
    "cosmo_hip_set_cones_ex": (C.c_int32, [C.c_void_p, C.c_int64, _PI32, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_set_custom_cone": (C.c_int32, [C.c_void_p, C.c_int64, PROJECT_FN, CONE_TEST_FN, CONE_TEST_FN, C.c_void_p]),
    "cosmo_hip_set_params": (C.c_int32, [C.c_void_p, C.POINTER(Params), _PR]),
    "cosmo_hip_update_rho": (C.c_int32, [C.c_void_p, _PR]),
    "cosmo_hip_set_scaling": (C.c_int32, [C.c_void_p, _PR, _PR, C.c_double]),
    "cosmo_hip_set_scaling_full": (C.c_int32, [C.c_void_p, _PR, _PR, _PR, _PR, C.c_double, C.c_double]),
    "cosmo_hip_default_accel_params": (None, [C.POINTER(AccelParams)]),
    "cosmo_hip_set_accelerator": (C.c_int32, [C.c_void_p, C.POINTER(AccelParams)]),
    "cosmo_hip_get_accel_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_get_accel_restarts": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_scale_ruiz": (C.c_int32, [C.c_void_p, C.c_int64, C.c_double, C.c_double, _PR, _PR, _PD]),
    "cosmo_hip_update_qb": (C.c_int32, [C.c_void_p, _PR, _PR]),
    "cosmo_hip_get_rho_classes": (C.c_int32, [C.c_void_p, _PI32]),
    "cosmo_hip_get_rho_vec": (C.c_int32, [C.c_void_p, _PR]),
    "cosmo_hip_kkt_solve": (C.c_int32, [C.c_void_p, _PR, _PR, _PI64]),
    "cosmo_hip_project": (C.c_int32, [C.c_void_p, _PR, _PI64, _PI32]),
    "cosmo_hip_spmv": (C.c_int32, [C.c_void_p, C.c_int32, _PR, _PR]),
    "cosmo_hip_set_iterates": (C.c_int32, [C.c_void_p, _PR, _PR, _PR]),
    "cosmo_hip_admm_init": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_admm_iterate": (C.c_int32, [C.c_void_p, C.c_int64]),
    "cosmo_hip_admm_iterate_checked": (C.c_int32, [C.c_void_p, C.c_int64, _PI32]),
    "cosmo_hip_residuals": (C.c_int32, [C.c_void_p, _PD]),
    "cosmo_hip_optimize": (C.c_int32, [C.c_void_p, C.POINTER(ResultStruct)]),
    "cosmo_hip_get_iterates": (C.c_int32, [C.c_void_p, _PR, _PR, _PR, _PR]),
    "cosmo_hip_cg_persist_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_fold_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_kkt_recurrence": (C.c_char_p, [C.c_void_p]),
    "cosmo_hip_polar_dataflow_stats": (C.c_int32, [C.c_void_p, _PD]),
    "cosmo_hip_polar_dataflow_reset_timing": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_get_kkt_solution": (C.c_int32, [C.c_void_p, _PR]),
    "cosmo_hip_get_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_set_setup_time": (C.c_int32, [C.c_void_p, C.c_double]),
    "cosmo_hip_get_rho_interval": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_time_spmv": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _PD, _PD]),
    "cosmo_hip_set_profiling": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cosmo_hip_get_kernel_times": (C.c_int32, [C.c_void_p, _PD, _PI64]),
    "cosmo_hip_kernel_class_name": (C.c_char_p, [C.c_int32]),
    "cosmo_hip_psd_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_set_psd_projection": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cosmo_hip_polar_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_polar_streamk_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_polar_schedule": (C.c_int32, [C.c_int32, _PD, _PI32]),
    "cosmo_hip_time_psd_product": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _PD, _PD]),
    "cosmo_hip_comm_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "cosmo_hip_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "cosmo_hip_comm_destroy": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_set_cone_shard": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_polar_depth_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_time_krylov": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "cosmo_hip_comm_selftest": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_comm_allreduce_check": (C.c_int32, [C.c_void_p, C.c_int64, _PI64]),
    "cosmo_hip_comm_init_hostshm": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    "cosmo_hip_comm_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_set_cone_ownership": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64]),
    "cosmo_hip_set_row_shard": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_row_shard_info": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_comm_stats_ex": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_create": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int64, C.c_int64]),
    "cosmo_hip_batch_destroy": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_batch_last_error": (C.c_char_p, [C.c_void_p]),
    "cosmo_hip_batch_set_problem": (C.c_int32, [C.c_void_p, C.c_int64, _PI64, _PI64, _PR, _PI64, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_batch_set_cones": (C.c_int32, [C.c_void_p, C.c_int64, _PI32, _PI64, _PR, _PR]),
    "cosmo_hip_batch_set_cones_ex": (C.c_int32, [C.c_void_p, C.c_int64, _PI32, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_batch_set_scaling": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, C.c_double]),
    "cosmo_hip_batch_set_params": (C.c_int32, [C.c_void_p, C.POINTER(Params)]),
    "cosmo_hip_batch_set_accelerator": (C.c_int32, [C.c_void_p, C.POINTER(AccelParams)]),
    "cosmo_hip_batch_get_accel_stats": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_get_rho_classes": (C.c_int32, [C.c_void_p, C.c_int64, _PI32]),
    "cosmo_hip_batch_set_iterates": (C.c_int32, [C.c_void_p, _PR, _PR, _PR]),
    "cosmo_hip_batch_optimize": (C.c_int32, [C.c_void_p, C.POINTER(ResultStruct)]),
    "cosmo_hip_batch_iterate": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32]),
    "cosmo_hip_batch_get_iterates": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, _PR, _PR]),
    "cosmo_hip_batch_get_counters": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_kernel_info": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_group_create": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64]),
    "cosmo_hip_batch_group_destroy": (C.c_int32, [C.c_void_p]),
    "cosmo_hip_batch_group_last_error": (C.c_char_p, [C.c_void_p]),
    "cosmo_hip_batch_group_set_problem": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, _PI64, _PI64, _PR, _PI64, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_batch_group_set_cones": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, _PI32, _PI64, _PR, _PR, _PR]),
    "cosmo_hip_batch_group_set_scaling": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, C.c_double]),
    "cosmo_hip_batch_group_set_scaling_full": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, _PR, _PR, C.c_double, C.c_double]),
    "cosmo_hip_batch_group_set_accelerator": (C.c_int32, [C.c_void_p, C.POINTER(AccelParams)]),
    "cosmo_hip_batch_group_set_params": (C.c_int32, [C.c_void_p, C.POINTER(Params)]),
    "cosmo_hip_batch_group_class_info": (C.c_int32, [C.c_void_p, _PI64, _PI64, _PI64]),
    "cosmo_hip_batch_group_run_info": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_group_get_rho_interval": (C.c_int32, [C.c_void_p, C.c_int64, _PI64]),
    "cosmo_hip_batch_group_set_iterates": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, _PR]),
    "cosmo_hip_batch_group_optimize": (C.c_int32, [C.c_void_p, C.POINTER(ResultStruct)]),
    "cosmo_hip_batch_group_get_iterates": (C.c_int32, [C.c_void_p, C.c_int64, _PR, _PR, _PR, _PR]),
    "cosmo_hip_batch_group_get_counters": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_hip_batch_group_get_accel_stats": (C.c_int32, [C.c_void_p, _PI64]),
}


def _is_f32(dtype):
    return np.dtype(dtype) == np.float32


def load_library(dtype=np.float64):
    """Loads libcosmo_hip.so (dtype float64) or libcosmo_hip_f32.so (float32), both built by `__graft_entry__.build()` /
    `make -C cosmo.jl_amd/csrc` from the same sources.  Raises if absent."""
    f32 = _is_f32(dtype)
    if f32 in _libs:
        return _libs[f32]
    path = LIB_PATH_F32 if f32 else LIB_PATH
    if not os.path.exists(path):
        raise ImportError("%s not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % (os.path.basename(path), path))
    try:
        # torch bundles its own libamdhip64; if it is going to live in this process (tests, bench: torch.distributed,
        # device selection) it must be loaded FIRST so that both share one HIP runtime.  Pure plumbing: no torch
        # symbol is used by the library, and the binding works without torch installed.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    preal = _PF if f32 else _PD
    fnmap = {PROJECT_FN: PROJECT_FN_F32, CONE_TEST_FN: CONE_TEST_FN_F32} if f32 else {}
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = [preal if a is _PR else fnmap.get(a, a) for a in args]
    got = int(lib.cosmo_hip_version())
    if got != ABI_VERSION:
        raise ImportError("%s reports ABI version %d, these bindings were generated for %d (include/cosmo_hip.h) -- rebuild the library "
                          "(`make -C cosmo.jl_amd/csrc`) or regenerate the bindings (`python tools/gen_abi_structs.py`)"
                          % (os.path.basename(path), got, ABI_VERSION))
    _libs[f32] = lib
    return lib


def _dp(a):
    """pointer to a data array in ITS element type (the arrays handed to a library are made with that library's dtype)"""
    return None if a is None else a.ctypes.data_as(_PF if a.dtype == np.float32 else _PD)


def _f64(a, n=None, name="array", dtype=np.float64):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if n is not None and a.size != n:
        raise ValueError("%s has length %d, expected %d" % (name, a.size, n))
    return a


def csc_julia(M, dtype=np.float64):
    """SciPy sparse -> the arrays of a Julia SparseMatrixCSC{T,Int64} (1-based, sorted rows), T = Float64 | Float32."""
    import scipy.sparse as sp
    # copy: a dtype conversion copies `data` but may SHARE `indices` with the caller's matrix; sorting the indices in place would
    # then permute the caller's index array without its values
    M = sp.csc_matrix(M, dtype=dtype, copy=True)
    M.sum_duplicates()
    M.sort_indices()
    colptr = np.ascontiguousarray(M.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(M.indices, dtype=np.int64) + 1
    nzval = np.ascontiguousarray(M.data, dtype=dtype)
    return colptr, rowval, nzval


def default_params(dtype=np.float64):
    p = Params()
    load_library(dtype).cosmo_hip_default_params(C.byref(p))
    return p


class Handle:
    """Thin object wrapper: one method per C entry point, NumPy arrays in and out."""

    def __init__(self, device=0, dtype=np.float64):
        self.dtype = np.dtype(np.float32 if _is_f32(dtype) else np.float64)      # element type of every data array (cosmo_hip_real)
        self.lib = load_library(self.dtype)
        self._h = C.c_void_p()
        rc = self.lib.cosmo_hip_create(C.byref(self._h), int(device))
        if rc != OK:
            raise CosmoHipError(rc, "cosmo_hip_create failed (no MI355X visible? this library has no CPU path)")
        self.n = self.m = 0
        self.ncones = 0
        self._callbacks = {}

    def _f(self, a, n=None, name="array"):
        return _f64(a, n, name, self.dtype)

    def close(self):
        if self._h:
            self.lib.cosmo_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            msg = self.lib.cosmo_hip_last_error(self._h)
            raise CosmoHipError(rc, msg.decode() if msg else "")

    # ---- problem ----------------------------------------------------------------------------------------
    def set_problem(self, P, q, A, b):
        m, n = A.shape
        pc, pr, pv = csc_julia(P, self.dtype)
        ac, ar, av = csc_julia(A, self.dtype)
        q = self._f(q, n, "q"); b = self._f(b, m, "b")
        self._chk(self.lib.cosmo_hip_set_problem(self._h, n, m, pc.ctypes.data_as(_PI64), pr.ctypes.data_as(_PI64), _dp(pv),
                                                 ac.ctypes.data_as(_PI64), ar.ctypes.data_as(_PI64), _dp(av), _dp(q), _dp(b)))
        self.n, self.m = n, m

    def set_cones(self, types, dims, box_l=None, box_u=None, cone_param=None):
        t = np.ascontiguousarray(types, dtype=np.int32)
        d = np.ascontiguousarray(dims, dtype=np.int64)
        bl = self._f(box_l); bu = self._f(box_u)
        if cone_param is None:
            self._chk(self.lib.cosmo_hip_set_cones(self._h, t.size, t.ctypes.data_as(_PI32), d.ctypes.data_as(_PI64), _dp(bl), _dp(bu)))
        else:
            cp = self._f(cone_param, t.size, "cone_param")
            self._chk(self.lib.cosmo_hip_set_cones_ex(self._h, t.size, t.ctypes.data_as(_PI32), d.ctypes.data_as(_PI64), _dp(bl), _dp(bu), _dp(cp)))
        self.ncones = t.size
        self._callbacks = {}

    def set_custom_cone(self, cone, project, in_dual=None, in_pol_recc=None):
        """cosmo_hip_set_custom_cone: `project(x)` projects the NumPy view x (the cone's slice) in place; `in_dual(x, tol)` /
        `in_pol_recc(x, tol)` return truth values (optional).  The ctypes thunks are kept alive on the handle."""
        def _view(ptr, dim):
            return np.ctypeslib.as_array(ptr, shape=(int(dim),)) if dim > 0 else np.zeros(0, dtype=self.dtype)

        def _proj(ptr, dim, _user):
            project(_view(ptr, dim))

        def _mk(fn):
            if fn is None:
                return test_t(0)
            return test_t(lambda ptr, dim, tol, _user: 1 if fn(_view(ptr, dim), float(tol)) else 0)

        f32 = self.dtype == np.float32
        proj_t, test_t = (PROJECT_FN_F32, CONE_TEST_FN_F32) if f32 else (PROJECT_FN, CONE_TEST_FN)
        thunks = (proj_t(_proj), _mk(in_dual), _mk(in_pol_recc))
        self._callbacks[int(cone)] = thunks
        self._chk(self.lib.cosmo_hip_set_custom_cone(self._h, int(cone), thunks[0], thunks[1], thunks[2], None))

    def default_params(self):
        return default_params(self.dtype)

    def set_params(self, params, rho_vec=None):
        rv = self._f(rho_vec, self.m, "rho_vec")
        self._chk(self.lib.cosmo_hip_set_params(self._h, C.byref(params), _dp(rv)))

    def update_rho(self, rho_vec):
        rv = self._f(rho_vec, self.m, "rho_vec")
        self._chk(self.lib.cosmo_hip_update_rho(self._h, _dp(rv)))

    def set_accelerator(self, kind=ACCEL_ANDERSON, mem=15, min_mem=3, safeguard=True, safeguard_tol=2.0, start_iter=2, start_accuracy=None):
        """`_make_accelerator!` (src/setup.jl:10-16); kind ACCEL_EMPTY removes it."""
        ap = AccelParams()
        self.lib.cosmo_hip_default_accel_params(C.byref(ap))
        ap.kind, ap.mem, ap.min_mem, ap.safeguard = int(kind), int(mem), int(min_mem), 1 if safeguard else 0
        ap.safeguard_tol, ap.start_iter = float(safeguard_tol), int(start_iter)
        if start_accuracy is not None:
            ap.start_accuracy = float(start_accuracy)
        self._chk(self.lib.cosmo_hip_set_accelerator(self._h, C.byref(ap)))

    def accel_restarts(self):
        """(restarts because the memory was full, restarts because rho was adapted) of the last optimize"""
        out = np.zeros(2, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_get_accel_restarts(self._h, out.ctypes.data_as(_PI64)))
        return int(out[0]), int(out[1])

    def set_psd_projection(self, mode):
        """0 = verified matrix-sign iteration above side 16 (default), 1 = eigendecomposition (Jacobi) at every side; after set_cones."""
        self._chk(self.lib.cosmo_hip_set_psd_projection(self._h, int(mode)))

    def accel_stats(self):
        out = np.zeros(6, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_get_accel_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(accelerated=int(out[0]), accepted=int(out[1]), declined=int(out[2]), restarts=int(out[3]), active=bool(out[4]),
                    safeguarding_iter=int(out[5]))

    def scale_ruiz(self, iterations, min_scaling=1e-4, max_scaling=1e4):
        """Device Ruiz equilibration of the resident (unscaled) problem; returns (D, E, c)."""
        D = np.empty(self.n, dtype=self.dtype); E = np.empty(self.m, dtype=self.dtype); c = C.c_double(1.0)
        self._chk(self.lib.cosmo_hip_scale_ruiz(self._h, int(iterations), float(min_scaling), float(max_scaling), _dp(D), _dp(E), C.byref(c)))
        return D, E, float(c.value)

    def set_scaling(self, Dinv, Einv, cinv):
        self._chk(self.lib.cosmo_hip_set_scaling(self._h, _dp(self._f(Dinv, self.n)), _dp(self._f(Einv, self.m)), float(cinv)))

    def set_scaling_full(self, D, Dinv, E, Einv, c, cinv):
        self._chk(self.lib.cosmo_hip_set_scaling_full(self._h, _dp(self._f(D, self.n)), _dp(self._f(Dinv, self.n)), _dp(self._f(E, self.m)),
                                                      _dp(self._f(Einv, self.m)), float(c), float(cinv)))

    def update_qb(self, q=None, b=None):
        self._chk(self.lib.cosmo_hip_update_qb(self._h, _dp(self._f(q, self.n)), _dp(self._f(b, self.m))))

    def get_rho_classes(self):
        out = np.empty(self.m, dtype=np.int32)
        self._chk(self.lib.cosmo_hip_get_rho_classes(self._h, out.ctypes.data_as(_PI32)))
        return out

    def get_rho_vec(self):
        out = np.empty(self.m, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_get_rho_vec(self._h, _dp(out)))
        return out

    # ---- fine-grained ---------------------------------------------------------------------------------------
    def kkt_solve(self, rhs):
        rhs = self._f(rhs, self.n + self.m, "rhs")
        lhs = np.empty(self.n + self.m, dtype=self.dtype)
        it = C.c_int64(0)
        self._chk(self.lib.cosmo_hip_kkt_solve(self._h, _dp(lhs), _dp(rhs), C.byref(it)))
        return lhs, it.value

    def project(self, s):
        s = np.array(s, dtype=self.dtype).copy()
        ranks = np.empty(max(self.ncones, 1), dtype=np.int64)
        br = np.empty(max(self.ncones, 1), dtype=np.int32)
        self._chk(self.lib.cosmo_hip_project(self._h, _dp(s), ranks.ctypes.data_as(_PI64), br.ctypes.data_as(_PI32)))
        return s, ranks[:self.ncones], br[:self.ncones]

    def spmv(self, which, x):
        nin = {MAT_A: self.n, MAT_AT: self.m, MAT_P: self.n}[which]
        nout = {MAT_A: self.m, MAT_AT: self.n, MAT_P: self.n}[which]
        x = self._f(x, nin, "x")
        y = np.empty(nout, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_spmv(self._h, which, _dp(y), _dp(x)))
        return y

    # ---- loop -------------------------------------------------------------------------------------------------
    def set_iterates(self, x0=None, s0=None, mu0=None):
        self._chk(self.lib.cosmo_hip_set_iterates(self._h, _dp(self._f(x0, self.n)), _dp(self._f(s0, self.m)), _dp(self._f(mu0, self.m))))

    def admm_init(self):
        self._chk(self.lib.cosmo_hip_admm_init(self._h))

    def admm_iterate(self, n_iters):
        self._chk(self.lib.cosmo_hip_admm_iterate(self._h, int(n_iters)))

    def admm_iterate_checked(self, n_iters):
        st = C.c_int32(0)
        self._chk(self.lib.cosmo_hip_admm_iterate_checked(self._h, int(n_iters), C.byref(st)))
        return st.value

    def set_setup_time(self, seconds):
        self._chk(self.lib.cosmo_hip_set_setup_time(self._h, float(seconds)))

    def rho_interval(self):
        """(adaptive_rho_interval in force, iteration at which the automatic rule fixed it or -1)"""
        out = np.zeros(2, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_get_rho_interval(self._h, out.ctypes.data_as(_PI64)))
        return int(out[0]), int(out[1])

    def residuals(self):
        out = np.empty(5)
        self._chk(self.lib.cosmo_hip_residuals(self._h, _dp(out)))
        return out

    def optimize(self):
        r = ResultStruct()
        self._chk(self.lib.cosmo_hip_optimize(self._h, C.byref(r)))
        return r

    def get_iterates(self):
        N = self.n + self.m
        w = np.empty(N, dtype=self.dtype); wp = np.empty(N, dtype=self.dtype); s = np.empty(self.m, dtype=self.dtype); mu = np.empty(self.m, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_get_iterates(self._h, _dp(w), _dp(wp), _dp(s), _dp(mu)))
        return w, wp, s, mu

    def get_kkt_solution(self):
        sol = np.empty(self.n + self.m, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_get_kkt_solution(self._h, _dp(sol)))
        return sol

    def get_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_get_stats(self._h, out.ctypes.data_as(_PI64)))
        keys = ["admm_iters", "kkt_solves", "kkt_iters_total", "kkt_budget_stalls", "spmv_A", "spmv_AT", "spmv_P", "rho_updates"]
        return dict(zip(keys, out.tolist()))

    def cg_persist_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_cg_persist_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["enabled", "workgroups", "launches", "fallbacks", "tickets", "arrivals", "abort", "lds_per_quarter"], out.tolist()))

    def kkt_recurrence(self):
        """Which Krylov recurrence / kernels the KKT solves of this handle run (cosmo_hip_kkt_recurrence)."""
        return self.lib.cosmo_hip_kkt_recurrence(self._h).decode()

    def fold_stats(self):
        out = np.zeros(6, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_fold_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["enabled", "nnz", "terms", "tiles", "factored_rows", "stored_entries"], out.tolist()))

    # ---- measurement -------------------------------------------------------------------------------------------
    def time_spmv(self, which, reps=50):
        t = C.c_double(0); by = C.c_double(0)
        self._chk(self.lib.cosmo_hip_time_spmv(self._h, which, reps, C.byref(t), C.byref(by)))
        return t.value, by.value

    def psd_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_psd_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["max_sweeps_wg", "sweeps_large", "not_converged", "ncones"], out.tolist()))

    POLAR_STAT_KEYS = ["large_cones", "batch_cones", "tile_side", "k_split", "launches_64_1", "launches_96_1", "launches_96_2", "launches_batch",
                       "products_last_large", "fallback_rounds", "verified", "products_last_batch", "schedule_steps", "unverified", "projections",
                       "err_max_e18"]

    def polar_streamk_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_polar_streamk_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["enabled", "workgroups", "classes", "timeouts"], out.tolist()))

    def polar_stats(self):
        out = np.zeros(16, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_polar_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(self.POLAR_STAT_KEYS, out.tolist()))

    def polar_dataflow_stats(self, reset=False):
        """The batch's main schedule as one persistent dependency-driven launch (cosmo_hip_polar_dataflow_stats)."""
        out = np.zeros(8, dtype=np.float64)
        self._chk(self.lib.cosmo_hip_polar_dataflow_stats(self._h, out.ctypes.data_as(_PD)))
        if reset:
            self._chk(self.lib.cosmo_hip_polar_dataflow_reset_timing(self._h))
        return dict(enabled=int(out[0]), launches=int(out[1]), products_per_launch=int(out[2]), timed_launches=int(out[3]), avg_launch_seconds=float(out[4]),
                    flops_per_launch=float(out[5]), workgroups=int(out[6]), tiles_per_product=int(out[7]))

    def time_psd_product(self, which=0, reps=20):
        t = C.c_double(0); fl = C.c_double(0)
        self._chk(self.lib.cosmo_hip_time_psd_product(self._h, int(which), int(reps), C.byref(t), C.byref(fl)))
        return t.value, fl.value

    # ---- clique sharding -----------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = load_library().cosmo_hip_comm_unique_id(buf)
        if rc != OK:
            raise CosmoHipError(rc, "cosmo_hip_comm_unique_id failed")
        return bytes(buf)

    def comm_init(self, rank, nranks, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self.lib.cosmo_hip_comm_init(self._h, int(rank), int(nranks), buf))

    def set_cone_shard(self, first_cone):
        fc = np.ascontiguousarray(first_cone, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_set_cone_shard(self._h, fc.ctypes.data_as(_PI64)))

    def set_cone_ownership(self, lo, hi):
        self._chk(self.lib.cosmo_hip_set_cone_ownership(self._h, int(lo), int(hi)))

    def comm_selftest(self):
        self._chk(self.lib.cosmo_hip_comm_selftest(self._h))

    def polar_depth_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_polar_depth_stats(self._h, out.ctypes.data_as(_PI64)))
        d = dict(zip(["adaptive", "depth_min", "depth_max", "depth_mean_x1000", "weighted_products_x1000", "failed_verifications", "downward_probes", "projections"], out.tolist()))
        d["depth_mean"] = d.pop("depth_mean_x1000") / 1000.0
        d["weighted_products_per_projection"] = d.pop("weighted_products_x1000") / 1000.0
        return d

    def time_krylov(self, reps):
        """(seconds per Krylov iteration incl. the kernel boundaries between its launches, algorithmic bytes per iteration, launches per iteration)."""
        t, b, nl = C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        self._chk(self.lib.cosmo_hip_time_krylov(self._h, int(reps), C.byref(t), C.byref(b), C.byref(nl)))
        return t.value, b.value, nl.value

    def comm_allreduce_check(self, count):
        """Known-answer all-reduce of `count` reals through the loop's own exchange path (collective: all ranks call it).  The caller compares
        `hash` across the ranks: the replicated n-side of a row-sharded run needs the SAME bits everywhere."""
        out = np.zeros(6, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_comm_allreduce_check(self._h, int(count), out.ctypes.data_as(_PI64)))
        d = dict(zip(["exact_mismatches", "inexact_outside_bound", "hash", "transport", "nranks", "rccl_version_code"], out.tolist()))
        d["count"] = int(count)
        return d

    def comm_init_hostshm(self, rank, nranks, name: str):
        self._chk(self.lib.cosmo_hip_comm_init_hostshm(self._h, int(rank), int(nranks), name.encode()))

    def comm_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_comm_stats(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["nranks", "rank", "exchanges", "transport"], out.tolist()))

    def set_row_shard(self, first_cone):
        """Row sharding (csrc/rowshard.hip): this rank keeps the cones first_cone[rank] <= k < first_cone[rank+1] and their rows."""
        fc = np.ascontiguousarray(first_cone, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_set_row_shard(self._h, fc.ctypes.data_as(_PI64)))

    def row_shard_info(self):
        out = np.zeros(6, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_row_shard_info(self._h, out.ctypes.data_as(_PI64)))
        return dict(zip(["row_lo", "row_hi", "m_global", "nnz_A_local", "cones_local", "first_cone"], out.tolist()))

    def comm_stats_ex(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_comm_stats_ex(self._h, out.ctypes.data_as(_PI64)))
        d = dict(zip(["nranks", "rank", "collectives", "transport", "mode", "bytes", "allreduces", "allreduce_elems"], out.tolist()))
        d["mode"] = {0: "none", 1: "cones", 2: "rows"}[d["mode"]]
        return d

    def set_profiling(self, on):
        self._chk(self.lib.cosmo_hip_set_profiling(self._h, int(on)))

    def get_kernel_times(self):
        sec = np.zeros(NUM_KERNEL_CLASSES); cnt = np.zeros(NUM_KERNEL_CLASSES, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_get_kernel_times(self._h, _dp(sec), cnt.ctypes.data_as(_PI64)))
        out = {}
        for k in range(NUM_KERNEL_CLASSES):
            if cnt[k]:
                out[self.lib.cosmo_hip_kernel_class_name(k).decode()] = (float(sec[k]), int(cnt[k]))
        return out


class Batch:
    """Batch of independent problems with identical (n, m, cone structure): one persistent workgroup per problem."""

    def __init__(self, nprob, n, m, device=0, dtype=np.float64):
        self.dtype = np.dtype(np.float32 if _is_f32(dtype) else np.float64)
        self.lib = load_library(self.dtype)
        self._b = C.c_void_p()
        rc = self.lib.cosmo_hip_batch_create(C.byref(self._b), int(device), int(nprob), int(n), int(m))
        if rc != OK:
            raise CosmoHipError(rc, "cosmo_hip_batch_create failed (no MI355X visible? this library has no CPU path)")
        self.nprob, self.n, self.m = int(nprob), int(n), int(m)

    def _f(self, a, n=None, name="array"):
        return _f64(a, n, name, self.dtype)

    def close(self):
        if self._b:
            self.lib.cosmo_hip_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            msg = self.lib.cosmo_hip_batch_last_error(self._b)
            raise CosmoHipError(rc, msg.decode() if msg else "")

    def set_problem(self, k, P, q, A, b):
        pc, pr, pv = csc_julia(P, self.dtype)
        ac, ar, av = csc_julia(A, self.dtype)
        q = self._f(q, self.n, "q"); b = self._f(b, self.m, "b")
        self._chk(self.lib.cosmo_hip_batch_set_problem(self._b, int(k), pc.ctypes.data_as(_PI64), pr.ctypes.data_as(_PI64), _dp(pv),
                                                       ac.ctypes.data_as(_PI64), ar.ctypes.data_as(_PI64), _dp(av), _dp(q), _dp(b)))

    def set_cones(self, types, dims, box_l=None, box_u=None, cone_param=None):
        t = np.ascontiguousarray(types, dtype=np.int32); d = np.ascontiguousarray(dims, dtype=np.int64)
        bl = self._f(box_l); bu = self._f(box_u)
        if cone_param is None:
            self._chk(self.lib.cosmo_hip_batch_set_cones(self._b, t.size, t.ctypes.data_as(_PI32), d.ctypes.data_as(_PI64), _dp(bl), _dp(bu)))
        else:                                                 # alpha of the power cones (as Handle.set_cones)
            cp = self._f(cone_param, t.size)
            self._chk(self.lib.cosmo_hip_batch_set_cones_ex(self._b, t.size, t.ctypes.data_as(_PI32), d.ctypes.data_as(_PI64), _dp(bl), _dp(bu), _dp(cp)))

    def set_scaling(self, k, Dinv, Einv, cinv):
        self._chk(self.lib.cosmo_hip_batch_set_scaling(self._b, int(k), _dp(self._f(Dinv, self.n)), _dp(self._f(Einv, self.m)), float(cinv)))

    def set_params(self, params):
        self._chk(self.lib.cosmo_hip_batch_set_params(self._b, C.byref(params)))

    def set_accelerator(self, kind=ACCEL_ANDERSON, mem=15, min_mem=3, safeguard=True, safeguard_tol=2.0, start_iter=2, start_accuracy=None):
        """`_make_accelerator!` (src/setup.jl:10-16) for every problem of the batch; before set_params.  mem <= 16 in batch mode."""
        ap = AccelParams()
        self.lib.cosmo_hip_default_accel_params(C.byref(ap))
        ap.kind, ap.mem, ap.min_mem, ap.safeguard = int(kind), int(mem), int(min_mem), 1 if safeguard else 0
        ap.safeguard_tol, ap.start_iter = float(safeguard_tol), int(start_iter)
        if start_accuracy is not None:
            ap.start_accuracy = float(start_accuracy)
        self._chk(self.lib.cosmo_hip_batch_set_accelerator(self._b, C.byref(ap)))

    def accel_stats(self):
        """Per problem: dict of int64 arrays (accelerated, accepted, declined, restarts, active, safeguarding_iter)."""
        out = np.zeros(6 * self.nprob, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_get_accel_stats(self._b, out.ctypes.data_as(_PI64)))
        keys = ("accelerated", "accepted", "declined", "restarts", "active", "safeguarding_iter")
        return {k: out[i::6].copy() for i, k in enumerate(keys)}

    def get_rho_classes(self, k):
        out = np.empty(self.m, dtype=np.int32)
        self._chk(self.lib.cosmo_hip_batch_get_rho_classes(self._b, int(k), out.ctypes.data_as(_PI32)))
        return out

    def set_iterates(self, x0=None, s0=None, mu0=None):
        self._chk(self.lib.cosmo_hip_batch_set_iterates(self._b, _dp(self._f(x0, self.nprob * self.n)), _dp(self._f(s0, self.nprob * self.m)),
                                                        _dp(self._f(mu0, self.nprob * self.m))))

    def optimize(self):
        res = (ResultStruct * self.nprob)()
        self._chk(self.lib.cosmo_hip_batch_optimize(self._b, res))
        return list(res)

    def iterate(self, n_iters, with_init=False):
        """n_iters more loop bodies of every undecided problem (residual / adaptive-rho checks on schedule).  The infeasibility certificates
        are NOT evaluated here -- only optimize() cuts the persistent launch for them -- so rates measured through iterate() exclude their cost."""
        self._chk(self.lib.cosmo_hip_batch_iterate(self._b, int(n_iters), 1 if with_init else 0))

    def kernel_info(self):
        """Which kernel the batch runs: dict(form = 'streaming' | 'lds_image' | 'register_1_2' | 'register_2_4', sliced, lds_bytes, p_in_registers,
        registers / scratch_bytes per thread and static_lds_bytes of that instantiation (from the loaded code object), sorted_assignment)."""
        out = np.zeros(8, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_kernel_info(self._b, out.ctypes.data_as(_PI64)))
        return dict(form=("streaming", "lds_image", "register_1_2", "register_2_4")[int(out[0])], sliced=bool(out[1]), lds_bytes=int(out[2]), p_in_registers=bool(out[3]),
                    registers=int(out[4]), scratch_bytes=int(out[5]), static_lds_bytes=int(out[6]), sorted_assignment=bool(out[7] & 1), long_rows=bool(out[7] & 2))

    def counters(self):
        """Per problem: ADMM iterations, KKT solves, Krylov iterations in total (three int64 arrays)."""
        out = np.zeros(3 * self.nprob, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_get_counters(self._b, out.ctypes.data_as(_PI64)))
        return out[0::3].copy(), out[1::3].copy(), out[2::3].copy()

    def get_iterates(self, k):
        N = self.n + self.m
        w = np.empty(N, dtype=self.dtype); wp = np.empty(N, dtype=self.dtype); s = np.empty(self.m, dtype=self.dtype); mu = np.empty(self.m, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_batch_get_iterates(self._b, int(k), _dp(w), _dp(wp), _dp(s), _dp(mu)))
        return w, wp, s, mu


class BatchGroup:
    """Batch of independent problems of DIFFERENT structure (csrc/batch_group.hip): every problem brings its own (n, m, cones); the library
    partitions them into classes of identical structure, one `Batch` per class, and solves all classes concurrently.  The reference's batch is a
    loop over arbitrary models (src/solver.jl:78)."""

    def __init__(self, nprob, device=0, dtype=np.float64):
        self.dtype = np.dtype(np.float32 if _is_f32(dtype) else np.float64)
        self.lib = load_library(self.dtype)
        self._g = C.c_void_p()
        rc = self.lib.cosmo_hip_batch_group_create(C.byref(self._g), int(device), int(nprob))
        if rc != OK:
            raise CosmoHipError(rc, "cosmo_hip_batch_group_create failed (no MI355X visible? this library has no CPU path)")
        self.nprob = int(nprob)
        self.dims = [None] * self.nprob

    def _f(self, a, n=None, name="array"):
        return _f64(a, n, name, self.dtype)

    def close(self):
        if self._g:
            self.lib.cosmo_hip_batch_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            msg = self.lib.cosmo_hip_batch_group_last_error(self._g)
            raise CosmoHipError(rc, msg.decode() if msg else "")

    def set_problem(self, k, P, q, A, b):
        m, n = A.shape
        pc, pr, pv = csc_julia(P, self.dtype)
        ac, ar, av = csc_julia(A, self.dtype)
        q = self._f(q, n, "q"); b = self._f(b, m, "b")
        self._chk(self.lib.cosmo_hip_batch_group_set_problem(self._g, int(k), int(n), int(m), pc.ctypes.data_as(_PI64), pr.ctypes.data_as(_PI64), _dp(pv),
                                                             ac.ctypes.data_as(_PI64), ar.ctypes.data_as(_PI64), _dp(av), _dp(q), _dp(b)))
        self.dims[int(k)] = (int(n), int(m))

    def set_cones(self, k, types, dims, box_l=None, box_u=None, cone_param=None):
        t = np.ascontiguousarray(types, dtype=np.int32); d = np.ascontiguousarray(dims, dtype=np.int64)
        bl = self._f(box_l); bu = self._f(box_u)
        cp = self._f(cone_param, t.size) if cone_param is not None else None
        self._chk(self.lib.cosmo_hip_batch_group_set_cones(self._g, int(k), t.size, t.ctypes.data_as(_PI32), d.ctypes.data_as(_PI64), _dp(bl), _dp(bu), _dp(cp)))

    def set_scaling(self, k, Dinv, Einv, cinv):
        n, m = self.dims[int(k)]
        self._chk(self.lib.cosmo_hip_batch_group_set_scaling(self._g, int(k), _dp(self._f(Dinv, n)), _dp(self._f(Einv, m)), float(cinv)))

    def set_scaling_full(self, k, D, Dinv, E, Einv, c, cinv):
        n, m = self.dims[int(k)]
        self._chk(self.lib.cosmo_hip_batch_group_set_scaling_full(self._g, int(k), _dp(self._f(D, n)), _dp(self._f(Dinv, n)), _dp(self._f(E, m)), _dp(self._f(Einv, m)),
                                                                  float(c), float(cinv)))

    def set_accelerator(self, kind=ACCEL_ANDERSON, mem=15, min_mem=3, safeguard=True, safeguard_tol=2.0, start_iter=2, start_accuracy=None):
        ap = AccelParams()
        self.lib.cosmo_hip_default_accel_params(C.byref(ap))
        ap.kind, ap.mem, ap.min_mem, ap.safeguard = int(kind), int(mem), int(min_mem), 1 if safeguard else 0
        ap.safeguard_tol, ap.start_iter = float(safeguard_tol), int(start_iter)
        if start_accuracy is not None:
            ap.start_accuracy = float(start_accuracy)
        self._chk(self.lib.cosmo_hip_batch_group_set_accelerator(self._g, C.byref(ap)))

    def set_params(self, params):
        self._chk(self.lib.cosmo_hip_batch_group_set_params(self._g, C.byref(params)))

    def class_info(self, with_modes=False):
        """(number of structure classes, class index of every problem[, mode of every problem: 0 persistent batch kernel, 1 its own handle])"""
        nc = C.c_int64(0)
        cls = np.zeros(self.nprob, dtype=np.int64); mode = np.zeros(self.nprob, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_group_class_info(self._g, C.byref(nc), cls.ctypes.data_as(_PI64), mode.ctypes.data_as(_PI64)))
        return (int(nc.value), cls, mode) if with_modes else (int(nc.value), cls)

    def rho_interval(self, k):
        """(adaptive_rho_interval in force for problem k, iteration at which the automatic rule fixed it or -1)"""
        out = np.zeros(2, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_group_get_rho_interval(self._g, int(k), out.ctypes.data_as(_PI64)))
        return int(out[0]), int(out[1])

    def run_info(self):
        """worker threads and jobs of the last optimize (a bounded pool), classes, problems"""
        out = np.zeros(5, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_group_run_info(self._g, out.ctypes.data_as(_PI64)))
        return dict(zip(["workers", "jobs", "classes", "problems", "merged_classes"], out.tolist()))

    def set_iterates(self, k, x0=None, s0=None, mu0=None):
        n, m = self.dims[int(k)]
        self._chk(self.lib.cosmo_hip_batch_group_set_iterates(self._g, int(k), _dp(self._f(x0, n)), _dp(self._f(s0, m)), _dp(self._f(mu0, m))))

    def optimize(self):
        res = (ResultStruct * self.nprob)()
        self._chk(self.lib.cosmo_hip_batch_group_optimize(self._g, res))
        return list(res)

    def get_iterates(self, k):
        n, m = self.dims[int(k)]
        w = np.empty(n + m, dtype=self.dtype); wp = np.empty(n + m, dtype=self.dtype); s = np.empty(m, dtype=self.dtype); mu = np.empty(m, dtype=self.dtype)
        self._chk(self.lib.cosmo_hip_batch_group_get_iterates(self._g, int(k), _dp(w), _dp(wp), _dp(s), _dp(mu)))
        return w, wp, s, mu

    def counters(self):
        out = np.zeros(3 * self.nprob, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_group_get_counters(self._g, out.ctypes.data_as(_PI64)))
        return out[0::3].copy(), out[1::3].copy(), out[2::3].copy()

    def accel_stats(self):
        out = np.zeros(6 * self.nprob, dtype=np.int64)
        self._chk(self.lib.cosmo_hip_batch_group_get_accel_stats(self._g, out.ctypes.data_as(_PI64)))
        keys = ("accelerated", "accepted", "declined", "restarts", "active", "safeguarding_iter")
        return {k: out[i::6].copy() for i, k in enumerate(keys)}
