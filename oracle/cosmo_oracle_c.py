"""ctypes loader of the compiled C restatement of the ADMM loop (oracle/cosmo_oracle_c.c).

TEST / BASELINE INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Setup (Ruiz scaling, constraint classification, rho vector: src/setup.jl:18-64) is done by the NumPy oracle's `Workspace`;
`run(ws)` then executes the loop of src/solver.jl:137-176 in compiled C on that workspace's scaled data.  Supported: ZeroSet,
Nonnegatives, Box, SecondOrderCone, PsdCone and PsdConeTriangle, CG reduced KKT solver -- all five BASELINE configs -- optionally with the
reference's default AndersonAccelerator + safeguarding (Settings(accelerator="anderson"); no infeasibility certificates in this loop).
The PSD projections call LAPACK ?syevr / BLAS ?syrk (src/convexset.jl:163-189, 243-263) through the function pointers SciPy exports for
its bundled OpenBLAS (scipy.linalg.cython_lapack / cython_blas); threadpoolctl limits that library's threads as it would Julia's BLAS.
"""
import ctypes as C
import os
import numpy as np
from . import cosmo_oracle as O

_LIB = {}


class Params(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("sigma", "alpha", "rho", "eps_abs", "eps_rel", "tol_constant", "tol_exponent", "rho_min", "rho_max",
                                           "rho_eq_over_rho_ineq", "adaptive_rho_tolerance", "cinv")] + \
               [("max_iter", C.c_int64), ("adaptive_rho_max_adaptions", C.c_int64), ("check_termination", C.c_int32), ("adaptive_rho", C.c_int32),
                ("adaptive_rho_interval", C.c_int32), ("unscale", C.c_int32),
                ("accel", C.c_int32), ("acc_mem", C.c_int32), ("acc_min_mem", C.c_int32), ("safeguard", C.c_int32), ("acc_start_iter", C.c_int64),
                ("safeguard_tol", C.c_double)]


class CResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_rho_updates", C.c_int32), ("iter", C.c_int64), ("cg_iters_total", C.c_int64)] + \
               [(k, C.c_double) for k in ("cost", "r_prim", "r_dual", "max_norm_prim", "max_norm_dual", "rho", "iter_time")] + \
               [("safeguarding_iter", C.c_int64), ("num_accelerated", C.c_int64)]


STATUS = {0: "Undetermined", 1: "Solved", 2: "Max_iter_reached", 3: "Unsolved"}


def lib(native=False, f32=False):
    """native=True: the -march=native build (`make -C oracle native`), made by bench.py on the host whose cores it times.
    f32=True: the Float32 instantiation of the same file (-DOC_FLOAT), the checker of libcosmo_hip_f32.so."""
    native = "f32" if f32 else native
    if native not in _LIB:
        name = "libcosmo_oracle_c_f32.so" if f32 else ("libcosmo_oracle_c_native.so" if native else "libcosmo_oracle_c.so")
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", name)
        if not os.path.exists(path):
            raise RuntimeError("compiled oracle missing: run `make -C oracle%s` (or __graft_entry__.build())" % (" native" if native else ""))
        L = C.CDLL(path)
        L.cosmo_oracle_c_run.restype = C.c_int32
        L.cosmo_oracle_c_run_cones.restype = C.c_int32
        _LIB[native] = L
    return _LIB[native]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_FN = {}


def _blas_pointer(module, name):
    """Address of a Fortran-interface LAPACK / BLAS routine of SciPy's bundled OpenBLAS (the PyCapsule of its Cython API)."""
    key = (module, name)
    if key not in _FN:
        import importlib
        cap = importlib.import_module("scipy.linalg." + module).__pyx_capi__[name]
        get_name = C.pythonapi.PyCapsule_GetName; get_name.restype = C.c_char_p; get_name.argtypes = [C.py_object]
        get_ptr = C.pythonapi.PyCapsule_GetPointer; get_ptr.restype = C.c_void_p; get_ptr.argtypes = [C.py_object, C.c_char_p]
        _FN[key] = C.c_void_p(get_ptr(cap, get_name(cap)))
    return _FN[key]


CK_SOC, CK_PSD_TRIANGLE, CK_PSD_SQUARE = 1, 2, 3


def run(ws: "O.Workspace", native=False, dtype=np.float64):
    """Run the loop on a set-up NumPy-oracle workspace (not yet optimised).  Returns a dict with scaled and unscaled iterates.
    dtype=np.float32: the workspace's (Float64-scaled) data are rounded to Float32 and the loop runs in the Float32 build -- the same
    arrays, rounded the same way, are what libcosmo_hip_f32.so receives in the Float32 parity tests."""
    f32 = np.dtype(dtype) == np.float32
    T = np.float32 if f32 else np.float64
    CT = C.c_float if f32 else C.c_double
    st = ws.st
    assert st.kkt_solver.lower() == "cg"
    accel = ws.accelerator is not None                     # the reference's default AndersonAccelerator (Immediate / IterActivation; no certificates in this loop)
    assert not accel or st.acc_start_accuracy is None
    n, m = ws.n, ws.m
    kind = np.zeros(m, np.int32); bl = np.zeros(m, T); bu = np.zeros(m, T)
    off = 0
    ck, co, cd, cone_index = [], [], [], []
    for ic, c in enumerate(ws.cones):
        d = c.dim
        if c.kind == O.ZERO:
            kind[off:off + d] = 1
        elif c.kind == O.NONNEG:
            kind[off:off + d] = 2
        elif c.kind == O.BOX:
            kind[off:off + d] = 3; bl[off:off + d] = c.l; bu[off:off + d] = c.u
        elif c.kind in (O.SOC, O.PSD_TRIANGLE, O.PSD_SQUARE):          # slice cones: projected in the serial cone loop
            ck.append({O.SOC: CK_SOC, O.PSD_TRIANGLE: CK_PSD_TRIANGLE, O.PSD_SQUARE: CK_PSD_SQUARE}[c.kind]); co.append(off); cd.append(d); cone_index.append(ic)
        else:
            raise ValueError("C oracle: unsupported cone kind %r" % (c.kind,))
        off += d
    ckind = np.asarray(ck, np.int32); coff = np.asarray(co, np.int64); cdim = np.asarray(cd, np.int64)
    has_psd = bool(np.any(ckind != CK_SOC)) if len(ck) else False
    syevr = _blas_pointer("cython_lapack", "ssyevr" if f32 else "dsyevr") if has_psd else C.c_void_p(None)
    syrk = _blas_pointer("cython_blas", "ssyrk" if f32 else "dsyrk") if has_psd else C.c_void_p(None)
    rank_out = np.full(max(len(ck), 1), -1, np.int64); branch_out = np.full(max(len(ck), 1), -1, np.int32)
    proj_time = C.c_double(0.0)
    P, A = ws.P, ws.A
    Pp = P.indptr.astype(np.int64); Pi = P.indices.astype(np.int64); Px = np.ascontiguousarray(P.data, T)
    Ap = A.indptr.astype(np.int64); Ai = A.indices.astype(np.int64); Ax = np.ascontiguousarray(A.data, T)
    prm = Params(st.sigma, st.alpha, ws.rho, st.eps_abs, st.eps_rel, st.tol_constant, st.tol_exponent, st.RHO_MIN, st.RHO_MAX,
                 st.RHO_EQ_OVER_RHO_INEQ, st.adaptive_rho_tolerance, ws.sm.cinv, st.max_iter, st.adaptive_rho_max_adaptions,
                 st.check_termination, int(bool(st.adaptive_rho)), st.adaptive_rho_interval, int(st.scaling != 0),
                 int(accel), int(getattr(st, "acc_mem", 15)), int(getattr(st, "acc_min_mem", 3)), int(bool(st.safeguard)), int(st.acc_start_iter), float(st.safeguard_tol))
    x = ws.x.astype(T); s = ws.s.astype(T); mu = ws.mu.astype(T)
    cap = 64
    rho_updates = np.zeros(cap, T)
    res = CResult()
    cls = np.ascontiguousarray(ws.rho_class, np.int32)
    q = np.ascontiguousarray(ws.q, T); b = np.ascontiguousarray(ws.b, T)
    Dinv = np.ascontiguousarray(ws.sm.Dinv, T); Einv = np.ascontiguousarray(ws.sm.Einv, T)
    rho0 = np.ascontiguousarray(ws.rho_vec, T)
    rc = lib(native, f32).cosmo_oracle_c_run_cones(C.c_int64(n), C.c_int64(m), _p(Pp, C.c_int64), _p(Pi, C.c_int64), _p(Px, CT), _p(Ap, C.c_int64),
                                  _p(Ai, C.c_int64), _p(Ax, CT), _p(q, CT), _p(b, CT), _p(Dinv, CT),
                                  _p(Einv, CT), _p(cls, C.c_int32), _p(kind, C.c_int32), _p(bl, CT), _p(bu, CT),
                                  C.byref(prm), _p(rho0, CT), _p(x, CT), _p(s, CT), _p(mu, CT),
                                  _p(rho_updates, CT), C.c_int32(cap), C.byref(res),
                                  C.c_int64(len(ck)), _p(ckind, C.c_int32), _p(coff, C.c_int64), _p(cdim, C.c_int64), syevr, syrk,
                                  _p(rank_out, C.c_int64), _p(branch_out, C.c_int32), C.byref(proj_time))
    if rc != 0:
        raise MemoryError("cosmo_oracle_c_run_cones failed (%d)" % rc)
    out = dict(status=STATUS[res.status], iter=int(res.iter), safeguarding_iter=int(res.safeguarding_iter), num_accelerated=int(res.num_accelerated),
               cg_iters_total=int(res.cg_iters_total), obj_val=res.cost, r_prim=res.r_prim,
               r_dual=res.r_dual, max_norm_prim=res.max_norm_prim, max_norm_dual=res.max_norm_dual, iter_time=res.iter_time, proj_time=proj_time.value,
               rho_updates=list(rho_updates[:min(cap, res.n_rho_updates)]), x_scaled=x, s_scaled=s, mu_scaled=mu,
               psd_rank={ic: int(rank_out[j]) for j, ic in enumerate(cone_index) if ck[j] != CK_SOC},
               soc_branch={ic: int(branch_out[j]) for j, ic in enumerate(cone_index) if ck[j] == CK_SOC})
    if st.scaling != 0:                                            # reverse_scaling! (src/scaling.jl:170-179)
        out["x"] = ws.sm.D * x; out["s"] = ws.sm.Einv * s; out["y"] = -((ws.sm.E * mu) * ws.sm.cinv)
    else:
        out["x"], out["s"], out["y"] = x, s, -mu
    return out
