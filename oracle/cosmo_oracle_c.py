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
        L.cosmo_oracle_c_run_cones_direct.restype = C.c_int32
        L.cosmo_oracle_c_ldl_nnz.restype = C.c_int64
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


# ---- the reference's DEFAULT KKT solver on the CPU: QdldlKKTSolver (src/linear_solver/kktsolver.jl:285-320) ------------------------------------
def kkt_full(ws):
    """K = [P + sigma I, A'; A, -diag(1 ./ rho)] (assemble_kkt_triangle, kktsolver.jl:175-250; here both triangles)."""
    import scipy.sparse as sp
    n, m = ws.n, ws.m
    return sp.bmat([[ws.P + ws.st.sigma * sp.identity(n), ws.A.T], [ws.A, sp.diags(-1.0 / ws.rho_vec)]], format="csc")


def _pattern_key(P, A):
    import hashlib
    h = hashlib.sha256()
    for M in (P, A):
        M = M.tocsc()
        h.update(np.asarray(M.shape, np.int64).tobytes()); h.update(M.indptr.astype(np.int64).tobytes()); h.update(M.indices.astype(np.int64).tobytes())
    return h.hexdigest()[:20]


PERM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kkt_perm")


def kkt_ordering(ws, cache=True):
    """A fill-reducing symmetric permutation of the KKT matrix, standing in for AMD.jl's `amd(K)` inside `qdldl(K)` (no AMD in this image): the rows
    of A with at most one entry are degree <= 1 nodes of K's graph -- any minimum-degree rule takes them first, they create no fill -- and what is left
    ([x ; the other rows]) is ordered by SuperLU's multiple-minimum-degree ordering on the pattern of its Schur complement (scipy `splu(permc_spec=
    "MMD_AT_PLUS_A")`; SuperLU factorises numerically to hand the permutation out, ~1 min for BASELINE config 5, hence the cache under oracle/kkt_perm/
    keyed by the sparsity pattern -- orderings are setup work, outside iter_time on every side).  Returns perm with perm[k] = original index at position k."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    n, m = ws.n, ws.m
    key = _pattern_key(ws.P, ws.A)
    path = os.path.join(PERM_DIR, "perm_%s.npz" % key)
    Ar = ws.A.tocsr()
    rn = np.diff(Ar.indptr)
    single = np.where(rn <= 1)[0]; multi = np.where(rn > 1)[0]
    nred = n + len(multi)
    old_of_new = None
    if cache and os.path.exists(path):                                                    # (the file holds the ordering of the reduced graph only)
        cand = np.load(path)["old_of_new"].astype(np.int64)
        if cand.size == nred and np.array_equal(np.sort(cand), np.arange(nred)):
            old_of_new = cand
    if old_of_new is None:
        As, Am = Ar[single], Ar[multi]
        S = (abs(ws.P) + (n + m + 1.0) * sp.identity(n) + abs(As).T @ abs(As)).tocsc()      # pattern of the Schur complement onto [x ; multi rows]
        Kr = sp.bmat([[S, abs(Am).T], [abs(Am), -(n + m + 1.0) * sp.identity(len(multi))]], format="csc") if len(multi) else S
        lu = sla.splu(Kr.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))   # (values are harmless: only the pattern matters)
        old_of_new = np.argsort(np.asarray(lu.perm_c, np.int64))                           # scipy: original column i sits at position perm_c[i]
        if cache:
            try:
                os.makedirs(PERM_DIR, exist_ok=True)
                np.savez_compressed(path, old_of_new=old_of_new.astype(np.int32))
            except OSError:
                pass
    red_index = np.concatenate([np.arange(n, dtype=np.int64), n + multi.astype(np.int64)])
    return np.concatenate([n + single.astype(np.int64), red_index[old_of_new]])


def kkt_permuted_upper(ws, perm):
    """(Kp, Ki, Kx, rho_pos): upper triangle of K[perm, perm] in CSC with ascending row indices (the diagonal is the LAST entry of every column) and
    the positions of the -1/rho_i diagonal entries (what update_rho! rewrites, kktsolver.jl:316-320)."""
    import scipy.sparse as sp
    n, m = ws.n, ws.m
    K = kkt_full(ws)
    Kp = sp.triu(K[perm][:, perm], format="csc")
    Kp.sort_indices()
    inv = np.empty(n + m, np.int64); inv[perm] = np.arange(n + m)
    rho_pos = Kp.indptr[inv[n:] + 1].astype(np.int64) - 1
    assert np.array_equal(Kp.indices[Kp.indptr[1:] - 1], np.arange(n + m)), "K needs a full diagonal"
    return Kp.indptr.astype(np.int64), Kp.indices.astype(np.int64), Kp.data.astype(np.float64), rho_pos


def ldl_nnz(ws, perm, cap=0, native=False):
    """nnz(L) of QDLDL's factor of K[perm, perm] from the elimination-tree pass alone; -2 if it exceeds `cap` (> 0)."""
    Kp, Ki, _, _ = kkt_permuted_upper(ws, perm)
    return int(lib(native).cosmo_oracle_c_ldl_nnz(C.c_int64(ws.n + ws.m), _p(Kp, C.c_int64), _p(Ki, C.c_int64), C.c_int64(cap)))


def run(ws: "O.Workspace", native=False, dtype=np.float64, direct=None):
    """Run the loop on a set-up NumPy-oracle workspace (not yet optimised).  Returns a dict with scaled and unscaled iterates.
    dtype=np.float32: the workspace's (Float64-scaled) data are rounded to Float32 and the loop runs in the Float32 build -- the same
    arrays, rounded the same way, are what libcosmo_hip_f32.so receives in the Float32 parity tests.
    direct = dict(perm=..., nnz_cap=...): the KKT systems are solved by the restated QdldlKKTSolver (LDL' of the permuted KKT matrix, refactorised
    at every rho update) instead of the CG reduced solver; the result carries `ldl` = {nnz_L, factor_s, n_factor, solve_s, n_solve, setup_s}."""
    f32 = np.dtype(dtype) == np.float32
    T = np.float32 if f32 else np.float64
    CT = C.c_float if f32 else C.c_double
    st = ws.st
    assert st.kkt_solver.lower() == "cg" or direct is not None
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
    common = (C.c_int64(n), C.c_int64(m), _p(Pp, C.c_int64), _p(Pi, C.c_int64), _p(Px, CT), _p(Ap, C.c_int64),
              _p(Ai, C.c_int64), _p(Ax, CT), _p(q, CT), _p(b, CT), _p(Dinv, CT),
              _p(Einv, CT), _p(cls, C.c_int32), _p(kind, C.c_int32), _p(bl, CT), _p(bu, CT),
              C.byref(prm), _p(rho0, CT), _p(x, CT), _p(s, CT), _p(mu, CT),
              _p(rho_updates, CT), C.c_int32(cap), C.byref(res),
              C.c_int64(len(ck)), _p(ckind, C.c_int32), _p(coff, C.c_int64), _p(cdim, C.c_int64), syevr, syrk,
              _p(rank_out, C.c_int64), _p(branch_out, C.c_int32), C.byref(proj_time))
    ldl_stats = None
    if direct is None:
        rc = lib(native, f32).cosmo_oracle_c_run_cones(*common)
    else:
        perm = np.ascontiguousarray(direct["perm"], np.int64)
        Kp, Ki, Kx, rho_pos = kkt_permuted_upper(ws, perm)
        Kx = np.ascontiguousarray(Kx, T)
        ldl_stats = np.zeros(8, np.float64)
        rc = lib(native, f32).cosmo_oracle_c_run_cones_direct(*common, _p(Kp, C.c_int64), _p(Ki, C.c_int64), _p(Kx, CT), _p(perm, C.c_int64), _p(rho_pos, C.c_int64),
                                                              C.c_int64(int(direct.get("nnz_cap", 0))), _p(ldl_stats, C.c_double))
        if rc == 5:
            raise OverflowError("nnz(L) exceeds the cap of %d" % int(direct.get("nnz_cap", 0)))
        if rc == 7:
            raise ValueError("Objective function is not convex.")          # the reference's own error (kktsolver.jl:304)
    if rc != 0:
        raise MemoryError("cosmo_oracle_c_run_cones failed (%d)" % rc)
    out = dict(status=STATUS[res.status], iter=int(res.iter), safeguarding_iter=int(res.safeguarding_iter), num_accelerated=int(res.num_accelerated),
               cg_iters_total=int(res.cg_iters_total), obj_val=res.cost, r_prim=res.r_prim,
               r_dual=res.r_dual, max_norm_prim=res.max_norm_prim, max_norm_dual=res.max_norm_dual, iter_time=res.iter_time, proj_time=proj_time.value,
               rho_updates=list(rho_updates[:min(cap, res.n_rho_updates)]), x_scaled=x, s_scaled=s, mu_scaled=mu,
               psd_rank={ic: int(rank_out[j]) for j, ic in enumerate(cone_index) if ck[j] != CK_SOC},
               soc_branch={ic: int(branch_out[j]) for j, ic in enumerate(cone_index) if ck[j] == CK_SOC})
    if ldl_stats is not None:
        out["ldl"] = dict(nnz_L=int(ldl_stats[0]), factor_s=float(ldl_stats[1]), n_factor=int(ldl_stats[2]), solve_s=float(ldl_stats[3]), n_solve=int(ldl_stats[4]),
                          positive_D=int(ldl_stats[5]), setup_s=float(ldl_stats[6]))
    if st.scaling != 0:                                            # reverse_scaling! (src/scaling.jl:170-179)
        out["x"] = ws.sm.D * x; out["s"] = ws.sm.Einv * s; out["y"] = -((ws.sm.E * mu) * ws.sm.cinv)
    else:
        out["x"], out["s"], out["y"] = x, s, -mu
    return out
