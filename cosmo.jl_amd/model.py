"""Host-side mirror of the reference's model interface for the ADMM hot path.

In the real drop-in the Julia front-end stays unchanged: `COSMO.Model`, `assemble!`, `set!`, `setup!` (Ruiz
scaling, constraint classification) run in Julia and `optimize!` hands the scaled problem to libcosmo_hip through
`ccall` (julia/CosmoHIP.jl).  Julia is not available in this image, so this module restates that thin host layer
in Python with the same names, argument meaning and error behaviour, so that the parity tests read like the
reference's tests:

    model = Model()
    assemble(model, P, q, [Constraint(A, b, Nonnegatives)], settings=Settings(kkt_solver=CGIndirectKKTSolver))
    res = optimize(model)          # -> Result(x, y, s, obj_val, iter, status, info, times)

Everything numerical inside the loop runs on the GPU; the only host arithmetic here is the one-off setup
(`scale_ruiz!`, src/scaling.jl:21-116) and the epilogue (`reverse_scaling!`, src/scaling.jl:170-179).
References are to /root/reference/src.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np
import scipy.sparse as sp

from . import _ffi

# ---- AbstractKKTSolver plugin names (src/linear_solver/kktsolver_indirect.jl:173-185) ----------------------------
CGIndirectKKTSolver = "CGIndirectKKTSolver"
MINRESIndirectKKTSolver = "MINRESIndirectKKTSolver"
IndirectReducedKKTSolverMINRES = "IndirectReducedKKTSolver(:MINRES)"
QdldlKKTSolver = "QdldlKKTSolver"
# opt-in, no reference counterpart: the same reduced system solved by single-reduction (Chronopoulos-Gear) CG, csrc/cg_sr.hip
CGSingleReductionKKTSolver = "CGSingleReductionKKTSolver"
# opt-in, no reference counterpart: Jacobi-preconditioned CG (IterativeSolvers' PCGIterable, Pl = diag of the reduced operator) on the assembled operator, csrc/cg_fold.hip
CGJacobiKKTSolver = "CGJacobiKKTSolver"
_KKT_KIND = {CGIndirectKKTSolver: _ffi.KKT_CG, MINRESIndirectKKTSolver: _ffi.KKT_MINRES, CGSingleReductionKKTSolver: _ffi.KKT_CG_SR, CGJacobiKKTSolver: _ffi.KKT_CG_JACOBI,
             IndirectReducedKKTSolverMINRES: _ffi.KKT_MINRES_REDUCED}


@dataclass
class OptionsFactory:
    """`with_options(SolverType; kwargs...)` (src/settings.jl:3-17)."""
    solver: str
    kwargs: dict = field(default_factory=dict)


def with_options(solver, **kwargs):
    return OptionsFactory(solver, kwargs)


class NoMerge:
    """COSMO.NoMerge (src/chordal_decomposition/clique_merging.jl:39)"""
    code = 0


class ParentChildMerge:
    """COSMO.ParentChildMerge(t_fill = 8, t_size = 8) (clique_merging.jl:80-101)"""
    code = 1


class CliqueGraphMerge:
    """COSMO.CliqueGraphMerge(edge_weight = ComplexityWeight()) (clique_merging.jl:42-75), the reference's default"""
    code = 2


@dataclass
class Settings:
    """Numeric fields of `COSMO.Settings` read by the hot path (src/settings.jl:101-139).  `accelerator` defaults to the
    EmptyAccelerator; `AndersonAccelerator` (csrc/anderson.hip) is opt-in, see the note next to the field."""
    rho: float = 0.1
    sigma: float = 1e-6
    alpha: float = 1.6
    eps_abs: float = 1e-5
    eps_rel: float = 1e-5
    eps_prim_inf: float = 1e-4
    eps_dual_inf: float = 1e-4
    max_iter: int = 5000
    verbose: bool = False
    kkt_solver: Union[str, OptionsFactory] = CGIndirectKKTSolver   # reference default is QdldlKKTSolver (CPU)
    check_termination: int = 25
    check_infeasibility: int = 40
    scaling: int = 10
    MIN_SCALING: float = 1e-4
    MAX_SCALING: float = 1e4
    adaptive_rho: bool = True
    adaptive_rho_interval: int = 40
    adaptive_rho_tolerance: float = 5.0
    adaptive_rho_max_adaptions: int = 2 ** 62
    RHO_MIN: float = 1e-6
    RHO_MAX: float = 1e6
    RHO_TOL: float = 1e-4
    RHO_EQ_OVER_RHO_INEQ: float = 1e3
    COSMO_INFTY: float = 1e20
    time_limit: float = 0.0
    obj_true: float = float("nan")        # src/settings.jl:132-133: convergence additionally requires |obj_true - cost| <= obj_true_tol
    obj_true_tol: float = 1e-3
    device: int = 0
    device_scaling: bool = True      # run scale_ruiz! on the MI355X (cosmo_hip_scale_ruiz) instead of on the host
    # accelerator / safeguard / safeguard_tol (src/settings.jl:96-98,136-138).  NOTE: the reference's default is
    # AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory, NoRegularizer}(mem = 15); this mirror defaults to the
    # EmptyAccelerator because the Anderson path restates an external package (parity unpinned) -- opt in with
    # accelerator=AndersonAccelerator or with_options(AndersonAccelerator, mem=...).
    # chordal decomposition (src/settings.jl:129-135): host C++ front-end libcosmo_chordal.so (include/cosmo_chordal.h)
    decompose: bool = True
    complete_dual: bool = False
    compact_transformation: bool = True
    merge_strategy: object = CliqueGraphMerge      # or with_options(ParentChildMerge, t_fill=8, t_size=8) / NoMerge
    accelerator: object = None
    accelerator_activation: object = 2  # 2 = ImmediateActivation; int k = IterActivation(k); AccuracyActivation(eps)
    safeguard: bool = True
    safeguard_tol: float = 2.0
    # how PsdCone / PsdConeTriangle are projected on the device (not a field of COSMO.Settings): "sign" = the verified matrix-sign iteration above side 16
    # (default), "eigen" = the eigendecomposition-based projection of the reference (syevr! + rank-k update, src/convexset.jl:163-189, 243-263) at every
    # side by Jacobi eigensolvers -- exact nnz_lambda, several times slower (cosmo_hip_set_psd_projection)
    psd_projection: str = "sign"
    # optimize() of ONE small model through the batch path (not a field of COSMO.Settings): a single-problem handle runs an ADMM iteration as a chain of
    # dependent launches (~150 us per iteration whatever the size); the batch kernels keep a whole problem in one persistent workgroup (INTEGRATION.md,
    # "Small problems": 20 against 65 ms per solve on the reference's portfolio example).  True: optimize() takes that path when the model fits it (no
    # distributed run, no chordal decomposition to apply, a structure the batch kernels take, an image small enough for the CU's LDS) -- same statuses and
    # solutions to solver accuracy, not the same bits; the model then has no single-problem handle (model.handle stays None).
    persistent_kernel: bool = False
    # accepted for drop-in compatibility with COSMO.Settings (src/settings.jl:101-139); they do not touch the hot path:
    nearly_ratio: float = 100.0            # only read by the MOI wrapper (is_primal_nearly_feasible, src/MOI_wrapper.jl:558,587)
    adaptive_rho_fraction: float = 0.4     # only with adaptive_rho_interval = 0 (the automatic interval, solver.jl:244-256)
    verbose_timing: bool = False           # the device loop always reports iter_time / proj_time


# ---- AbstractConvexSet subtypes on the hot path (src/convexset.jl) ---------------------------------------------------
class AbstractConvexSet:
    kind = -1

    def __init__(self, dim):
        if dim < 0:
            raise ValueError("dimension must be nonnegative")     # DomainError in the reference
        self.dim = int(dim)


class ZeroSet(AbstractConvexSet):
    kind = _ffi.ZERO


class Nonnegatives(AbstractConvexSet):
    kind = _ffi.NONNEG


class SecondOrderCone(AbstractConvexSet):
    kind = _ffi.SOC


class PsdCone(AbstractConvexSet):
    kind = _ffi.PSD_SQUARE

    def __init__(self, dim):
        super().__init__(dim)
        r = math.isqrt(self.dim)
        if r * r != self.dim:
            raise ValueError("dimension must be a square")        # src/convexset.jl:280
        self.sqrt_dim = r


class PsdConeTriangle(AbstractConvexSet):
    kind = _ffi.PSD_TRIANGLE

    def __init__(self, dim):
        super().__init__(dim)
        self.sqrt_dim = (math.isqrt(1 + 8 * self.dim) - 1) // 2   # src/convexset.jl:372


class ComplexPsdConeTriangle(AbstractConvexSet):
    """`PsdConeTriangle{T, Complex{T}}(dim)`, dim = r^2 (src/convexset.jl:345-380): Hermitian r x r matrices stored as the svec of the
    real part followed by sqrt(2) * the imaginary parts of the strict upper triangle."""
    kind = _ffi.PSD_TRIANGLE_COMPLEX

    def __init__(self, dim):
        super().__init__(dim)
        r = math.isqrt(self.dim)
        if r * r != self.dim:
            raise ValueError("dimension must be a square")
        self.sqrt_dim = r


class ExponentialCone(AbstractConvexSet):
    """K_exp (src/convexset.jl:497-507); MAX_ITERS = 100 and EXP_TOL = 1e-8 are the reference defaults and fixed here."""
    kind = _ffi.EXP

    def __init__(self, dim=3):
        super().__init__(3)


class DualExponentialCone(ExponentialCone):
    kind = _ffi.DUAL_EXP


class PowerCone(AbstractConvexSet):
    """K_pow(alpha) (src/convexset.jl:607-618); MAX_ITERS = 20 and POW_TOL = 1e-8 are the reference defaults."""
    kind = _ffi.POW

    def __init__(self, alpha):
        if not (0.0 < alpha < 1.0):                               # DomainError (:614)
            raise ValueError("The exponent alpha of the power cone has to be in (0, 1).")
        super().__init__(3)
        self.alpha = float(alpha)


class DualPowerCone(PowerCone):
    kind = _ffi.DUAL_POW


class AbstractConvexCone(AbstractConvexSet):
    """User extension point: `COSMO.AbstractConvexCone{T}` (src/projections.jl:5, docs/src/literate/custom_cone.jl:9-17).
    Subclass it, keep `dim`, and define `project(self, x)` (in place on the NumPy view of the cone's slice); optionally
    `in_dual(self, x, tol)` and `in_pol_recc(self, x, tol)` to take part in infeasibility detection (:62-68).  The methods run
    on the host, called back by the device library once per iteration (cosmo_hip_set_custom_cone)."""
    kind = _ffi.CUSTOM

    def project(self, x):
        raise NotImplementedError("a custom cone must define project(x)")      # MethodError in the reference

    in_dual = None
    in_pol_recc = None


# src/convexset.jl:953-958 (the generic rectify_scaling! fall-back scalar-scales user cones as well)
_SCALAR_SCALED = (_ffi.SOC, _ffi.PSD_SQUARE, _ffi.PSD_TRIANGLE, _ffi.EXP, _ffi.DUAL_EXP, _ffi.POW, _ffi.DUAL_POW, _ffi.PSD_TRIANGLE_COMPLEX,
                  _ffi.CUSTOM)


class AccuracyActivation:
    """`AccuracyActivation(start_accuracy)` (src/accelerator_interface.jl:14-21): switch the accelerator on once the residuals
    satisfy eps_abs = eps_rel = start_accuracy at a termination check."""

    def __init__(self, start_accuracy):
        self.start_accuracy = float(start_accuracy)


class IterActivation(int):
    """`IterActivation(start_iter)` (src/accelerator_interface.jl:5-12)."""


class EmptyAccelerator:
    """CA.EmptyAccelerator"""


# type parameters of CA.AndersonAccelerator{T, BT, MT, RE} (src/printing.jl:83-97, docs/src/acceleration.md:5,23)
class QRDecomp:
    """CA.QRDecomp"""


class NormalEquations:
    """CA.NormalEquations"""


class Type1:
    """CA.Type1: eta solves (X' F) eta = X' f"""


class Type2:
    """CA.Type2{QRDecomp} (bare `Type2`) / CA.Type2{NormalEquations} (`Type2[NormalEquations]`): eta = argmin ||f - F eta||"""
    method = QRDecomp

    def __class_getitem__(cls, method):
        if method is QRDecomp:
            return cls
        if method is NormalEquations:
            return _Type2NormalEquations
        raise TypeError("Type2{...}: QRDecomp or NormalEquations")


class _Type2NormalEquations(Type2):
    method = NormalEquations


class RestartedMemory:
    """CA.RestartedMemory: the history is emptied when it is full"""


class RollingMemory:
    """CA.RollingMemory: the oldest column is replaced by the newest"""


class NoRegularizer:
    """CA.NoRegularizer (the only regulariser on the device)"""


class AndersonAccelerator:
    """AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory, NoRegularizer}, the reference's default (csrc/anderson.hip, and inside the persistent
    batch kernels).  `AndersonAccelerator[Type1, RollingMemory]` etc. -- Julia's `AndersonAccelerator{Float64, Type1, RollingMemory, NoRegularizer}`
    (docs/src/acceleration.md:23-26; a leading dtype and a trailing NoRegularizer are accepted and ignored) -- select the other broyden types
    (Type1, Type2[NormalEquations]) and memory types on single-problem handles; Type2{QRDecomp} with RollingMemory does not exist in the package either."""
    broyden, memory = Type2, RestartedMemory
    accel_kind = _ffi.ACCEL_ANDERSON
    _variants: dict = {}

    def __class_getitem__(cls, params):
        params = params if isinstance(params, tuple) else (params,)
        bt, mt = Type2, RestartedMemory
        for prm in params:
            if isinstance(prm, type) and issubclass(prm, (Type1, Type2)):
                bt = prm
            elif prm in (RestartedMemory, RollingMemory):
                mt = prm
            elif prm is NoRegularizer or prm in (float, np.float64, np.float32):
                pass
            else:
                raise TypeError("AndersonAccelerator{...}: unknown type parameter %r" % (prm,))
        if bt is Type2 and mt is RestartedMemory:
            return AndersonAccelerator
        if bt is Type2 and mt is RollingMemory:
            raise TypeError("AndersonAccelerator{Type2{QRDecomp}, RollingMemory}: a rolling memory needs Type1 or Type2{NormalEquations}")
        key = (bt, mt)
        if key not in cls._variants:
            kind = {(Type1, RestartedMemory): _ffi.ACCEL_ANDERSON_TYPE1_RESTARTED, (Type1, RollingMemory): _ffi.ACCEL_ANDERSON_TYPE1_ROLLING,
                    (_Type2NormalEquations, RestartedMemory): _ffi.ACCEL_ANDERSON_TYPE2NE_RESTARTED,
                    (_Type2NormalEquations, RollingMemory): _ffi.ACCEL_ANDERSON_TYPE2NE_ROLLING}[key]
            cls._variants[key] = type("AndersonAccelerator_%s_%s" % (bt.__name__.strip("_"), mt.__name__), (AndersonAccelerator,),
                                      dict(broyden=bt, memory=mt, accel_kind=kind))
        return cls._variants[key]


class Box(AbstractConvexSet):
    kind = _ffi.BOX

    def __init__(self, l, u):
        l = np.array(l, dtype=np.float64).copy()
        u = np.array(u, dtype=np.float64).copy()
        if l.shape != u.shape:
            raise ValueError("bounds must be same length")
        bad = np.nonzero(l > u)[0]
        if bad.size:                                              # src/convexset.jl:824-828
            raise ValueError("Box set: inconsistent lower/upper bounds specified at index i = %d" % (bad[0] + 1))
        super().__init__(l.size)
        self.l, self.u = l, u


class Constraint:
    """`Constraint(A, b, K)`: A x + b in K (src/constraint.jl:47-108).  `K` may be a set instance or a set type."""

    def __init__(self, A, b, convex_set, dim: int = 0, indices: Optional[range] = None):
        if not sp.issparse(A):
            A = np.asarray(A, dtype=np.float64)
            if A.ndim == 0:
                A = A.reshape(1, 1)                               # scalar A: 1 x 1 (src/constraint.jl:110-120)
            elif A.ndim == 1:
                A = A.reshape(-1, 1)                              # a Julia Vector is a column: n rows, one variable
        A = sp.csc_matrix(A, dtype=np.float64)
        b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=np.float64)
        if b.ndim > 1 and min(b.shape) != 1:
            raise ValueError("Input b must be a vector or a scalar.")                              # :62
        b = np.atleast_1d(b).ravel().copy()
        if A.shape[0] != b.size:
            raise ValueError("The dimensions of matrix A and vector b don't match.")               # :60
        if isinstance(convex_set, type):
            if issubclass(convex_set, PowerCone):                                                   # ArgumentCones (:97)
                raise TypeError("You can't create a constraint by passing the convex set as a type, if your convex set is a %s. "
                                "Please pass an object." % convex_set.__name__)
            nrows = A.shape[0]
            if convex_set is PsdConeTriangle and nrows != 1 and math.isqrt(nrows) ** 2 == nrows:
                convex_set = ComplexPsdConeTriangle(nrows)       # the reference deduces real / complex from the dimension (:101-106)
            else:
                convex_set = convex_set(nrows)
        if A.shape[0] != convex_set.dim:
            raise ValueError("The row dimension of A doesn't match the dimension of the constraint set.")   # :61
        if indices is not None:                                   # src/constraint.jl:63-70 (0-based half-open `range` here)
            idx = list(indices)
            if not idx or idx[0] < 0 or any(b2 - a2 != 1 for a2, b2 in zip(idx, idx[1:])):
                raise ValueError("The index range for x has to be increasing and nonnegative.")
            if dim < idx[-1] + 1:
                raise ValueError("The dimension of x: %d must be equal or higher than the the stop value of indices: %d." % (dim, idx[-1] + 1))
            if len(idx) != A.shape[1]:
                raise ValueError("The index range does not match the number of columns of A.")
            Ac = sp.lil_matrix((A.shape[0], dim))
            Ac[:, idx] = A
            A = Ac.tocsc()
        self.A, self.b, self.convex_set = A, b, convex_set


@dataclass
class ResultInfo:
    r_prim: float
    r_dual: float
    max_norm_prim: float
    max_norm_dual: float
    rho_updates: List[float]


@dataclass
class ResultTimes:
    solver_time: float = 0.0
    setup_time: float = 0.0
    iter_time: float = 0.0
    proj_time: float = 0.0


@dataclass
class Result:
    """`COSMO.Result` (src/types.jl:93-112)."""
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    obj_val: float
    iter: int
    status: str
    info: ResultInfo
    times: ResultTimes
    kkt_iters_total: int = 0
    safeguarding_iter: int = 0          # Result.safeguarding_iter (src/types.jl:93-112); `iter` includes them (src/solver.jl:196)


@dataclass
class ScaleMatrices:
    D: np.ndarray
    Dinv: np.ndarray
    E: np.ndarray
    Einv: np.ndarray
    c: float = 1.0
    cinv: float = 1.0


# ---- setup!: Ruiz equilibration (src/scaling.jl) ---------------------------------------------------------------------
def _limit(v, lo, hi):
    # limit_scaling! (src/scaling.jl:10-13): below MIN -> 1, above MAX -> MAX
    return np.where(v < lo, 1.0, np.where(v > hi, hi, v))


def _col_maxabs(M, out, reset):
    if reset:
        out[:] = 0.0
    if M.nnz:
        cols = np.repeat(np.arange(M.shape[1]), np.diff(M.indptr))
        np.maximum.at(out, cols, np.abs(M.data))


def _scale_csc(M, L, R):
    if M.nnz:
        cols = np.repeat(np.arange(M.shape[1]), np.diff(M.indptr))
        f = np.ones(M.nnz)
        if L is not None:
            f = L[M.indices]
            if R is not None:
                f = f * R[cols]
        elif R is not None:
            f = R[cols]
        M.data *= f


def scale_ruiz(P, q, A, b, sets: Sequence[AbstractConvexSet], st: Settings) -> ScaleMatrices:
    """`scale_ruiz!` (src/scaling.jl:21-116), in place on P, q, A, b and the Box bounds."""
    m, n = A.shape
    D = np.ones(n); E = np.ones(m); c = 1.0
    Dw = np.ones(n); Ew = np.ones(m)
    for _ in range(st.scaling):
        _col_maxabs(P, Dw, True); _col_maxabs(A, Dw, False)
        Ew[:] = 0.0
        if A.nnz:
            np.maximum.at(Ew, A.indices, np.abs(A.data))
        Dw[:] = 1.0 / np.sqrt(_limit(Dw, st.MIN_SCALING, st.MAX_SCALING))
        Ew[:] = 1.0 / np.sqrt(_limit(Ew, st.MIN_SCALING, st.MAX_SCALING))
        _scale_csc(P, Dw, Dw); _scale_csc(A, Ew, Dw)
        q *= Dw; b *= Ew; D *= Dw; E *= Ew
        _col_maxabs(P, Dw, True)
        mean_col = float(np.mean(Dw)) if n else 0.0
        nq = float(np.max(np.abs(q))) if n else 0.0
        if mean_col != 0.0 and nq != 0.0:
            nq = float(_limit(nq, st.MIN_SCALING, st.MAX_SCALING))
            sc = float(_limit(max(nq, mean_col), st.MIN_SCALING, st.MAX_SCALING))
            ct = 1.0 / sc
            P.data *= ct; q *= ct; c *= ct
    off = 0
    Ew[:] = 1.0
    changed = False
    for K in sets:                                                # rectify_set_scalings! (:129-142)
        if K.kind in _SCALAR_SCALED and K.dim > 0:
            Ew[off:off + K.dim] = float(np.mean(E[off:off + K.dim])) / E[off:off + K.dim]
            changed = True
        off += K.dim
    if changed:
        _scale_csc(A, Ew, None); b *= Ew; E *= Ew
    if (abs(P - P.T)).nnz != 0:                                    # symmetrize_full! (:99)
        Ps = ((P + P.T) / 2.0).tocsc(); Ps.sort_indices()
        P.data, P.indices, P.indptr = Ps.data, Ps.indices, Ps.indptr
    off = 0
    for K in sets:                                                # scale_sets! (:145-154)
        if K.kind == _ffi.BOX:
            K.l *= E[off:off + K.dim]; K.u *= E[off:off + K.dim]
        off += K.dim
    return ScaleMatrices(D, 1.0 / D, E, 1.0 / E, c, 1.0 / c)


_SORT = {_ffi.ZERO: 1, _ffi.NONNEG: 2, _ffi.BOX: 3, _ffi.SOC: 4, _ffi.PSD_SQUARE: 5, _ffi.PSD_TRIANGLE: 6,
         _ffi.EXP: 6, _ffi.DUAL_EXP: 6, _ffi.POW: 6, _ffi.DUAL_POW: 6, _ffi.PSD_TRIANGLE_COMPLEX: 6, _ffi.CUSTOM: 6}     # sort_sets fall-through (src/interface.jl:466-475)


def _copy_set(K):
    if K.kind == _ffi.CUSTOM:
        return K                                     # user object: shared, as the reference shares it with the Constraint
    if isinstance(K, Box):
        return Box(K.l, K.u)
    if isinstance(K, PowerCone):
        return type(K)(K.alpha)
    return type(K)(K.dim)


class Model:
    """`COSMO.Model` / `Workspace` (src/types.jl:348-403) as far as the hot path needs it."""

    def __init__(self, dtype=np.float64):
        # COSMO.Model{T}(), T = Float64 | Float32 (src/types.jl:348, :390-403): the device library of that element type runs the loop
        # (libcosmo_hip.so / libcosmo_hip_f32.so); this host mirror keeps its own copies in float64 and hands the arrays over in T
        self.dtype = np.dtype(np.float32 if np.dtype(dtype) == np.float32 else np.float64)
        self.P = self.q = self.A = self.b = None
        self.sets: List[AbstractConvexSet] = []
        self.settings = Settings()
        self.is_assembled = False
        self.is_scaled = False
        self.is_optimized = False
        self.sm: Optional[ScaleMatrices] = None
        self.handle: Optional[_ffi.Handle] = None
        self.chordal = None            # _chordal.Decomposition of a decomposed model (ws.ci)
        self.x = self.s = self.mu = None
        self.n = self.m = 0

    # set! (src/interface.jl:218-250): data already in internal sign convention (A x + s = b)
    def set(self, P, q, A, b, convex_sets: Sequence[AbstractConvexSet], settings: Optional[Settings] = None):
        def _mat(M):                                             # numbers, vectors, dense and sparse matrices (src/interface.jl:79-106)
            return M if sp.issparse(M) else np.atleast_2d(np.asarray(M, dtype=np.float64))

        def _vec(v):
            return np.array(v.toarray() if sp.issparse(v) else v, dtype=np.float64).ravel().copy()
        P = sp.csc_matrix(_mat(P), dtype=np.float64, copy=True); A = sp.csc_matrix(_mat(A), dtype=np.float64, copy=True)
        P.sort_indices(); A.sort_indices()
        q = _vec(q); b = _vec(b)
        n = q.size; m = b.size
        if P.shape != (n, n) or A.shape != (m, n):
            raise ValueError("The dimensions of P, q, A, b are inconsistent.")
        if sum(K.dim for K in convex_sets) != m:
            raise ValueError("The dimensions of the convex sets don't match the number of rows of A.")
        self.empty()
        self.P, self.q, self.A, self.b = P, q, A, b
        self.sets = [_copy_set(K) for K in convex_sets]
        self.n, self.m = n, m
        if settings is not None:
            self.settings = settings
        self.x = np.zeros(n); self.s = np.zeros(m); self.mu = np.zeros(m)
        self.is_assembled = True

    def empty(self):
        """`empty_model!` (src/interface.jl:98-114)."""
        if self.handle is not None:
            self.handle.close()
        self.handle = None
        self.is_assembled = self.is_scaled = self.is_optimized = False
        self.sm = None
        self.chordal = None


def convex_sets_from_dict(cone: dict, l=None, u=None) -> List[AbstractConvexSet]:
    """`convex_sets_from_dict` (src/interface.jl:312-367): the SCS-style cone dictionary of the cosmo-python interface --
    "f" ZeroSet, "l" Nonnegatives, "q" SecondOrderCones, "s" PsdConeTriangles, "ep" / "ed" (dual) exponential cones, "p" power cones
    (negative exponent = dual), "b" Box(l, u) -- in this fixed order."""
    sets: List[AbstractConvexSet] = []
    if "f" in cone:
        sets.append(ZeroSet(int(cone["f"])))
    if "l" in cone:
        sets.append(Nonnegatives(int(cone["l"])))
    for d in np.atleast_1d(cone.get("q", [])):
        sets.append(SecondOrderCone(int(d)))
    for d in np.atleast_1d(cone.get("s", [])):
        sets.append(PsdConeTriangle(int(d)))
    sets += [ExponentialCone() for _ in range(int(cone.get("ep", 0)))]
    sets += [DualExponentialCone() for _ in range(int(cone.get("ed", 0)))]
    for e in np.atleast_1d(cone.get("p", [])):
        sets.append(PowerCone(float(e)) if e >= 0 else DualPowerCone(-1.0 * float(e)))
    if "b" in cone:
        sets.append(Box(l, u))
    return sets


def set_csc(model: Model, P_rowval, P_colptr, P_nzval, q, A_rowval, A_colptr, A_nzval, b, cone: dict, l, u, m: int, n: int,
            settings: Optional[Settings] = None):
    """`set!(model, Prowval, Pcolptr, Pnzval, q, Arowval, Acolptr, Anzval, b, cone, l, u, m, n[, settings])`
    (src/interface.jl:252-301): raw CSC triplets (0-based here, as cosmo-python passes them before `juliafy_integers`) in the
    internal convention A x + s = b plus the cone dictionary."""
    P = sp.csc_matrix((np.asarray(P_nzval, dtype=np.float64), np.asarray(P_rowval, dtype=np.int64), np.asarray(P_colptr, dtype=np.int64)), shape=(n, n))
    A = sp.csc_matrix((np.asarray(A_nzval, dtype=np.float64), np.asarray(A_rowval, dtype=np.int64), np.asarray(A_colptr, dtype=np.int64)), shape=(m, n))
    model.set(P, q, A, b, convex_sets_from_dict(cone, l, u), settings)


def assemble(model: Model, P, q, constraints: Union[Constraint, Sequence[Constraint]], settings: Optional[Settings] = None,
             x0=None, y0=None):
    """`assemble!` (src/interface.jl:30-77): merge Zero/Nonnegatives constraints, stable-sort the sets
    Zero < Nonneg < Box < SOC < Psd < PsdTriangle, and store A := -A_c, b := b_c (:478-484)."""
    cons = [constraints] if isinstance(constraints, Constraint) else list(constraints)
    for kind, ctor in ((_ffi.ZERO, ZeroSet), (_ffi.NONNEG, Nonnegatives)):
        idx = [i for i, c in enumerate(cons) if c.convex_set.kind == kind]
        if len(idx) > 1:
            Am = sp.vstack([cons[i].A for i in idx], format="csc")
            bm = np.concatenate([cons[i].b for i in idx])
            cons = [c for i, c in enumerate(cons) if i not in idx]
            cons.append(Constraint(Am, bm, ctor(bm.size)))
    cons.sort(key=lambda c: _SORT[c.convex_set.kind])
    A = sp.vstack([-c.A for c in cons], format="csc")
    b = np.concatenate([c.b for c in cons])
    model.set(P, q, A, b, [c.convex_set for c in cons], settings)
    if x0 is not None:
        warm_start_primal(model, x0)
    if y0 is not None:
        warm_start_dual(model, y0)


def warm_start_primal(model: Model, x0):
    """`warm_start_primal!` with a full vector also warm starts s = b - A x (src/interface.jl:130-148)."""
    x0 = np.array(x0, dtype=np.float64)
    model.x[:] = x0
    if model.is_scaled and getattr(model, "device_scaled", False):
        # Ruiz scaling ran on the device (setup: device_scaling): the host keeps the UNSCALED P, A and the scaled b = E b0, so the
        # unscaled slack b0 - A x0 is Einv .* b - A x0 (the reference holds the scaled A here, src/interface.jl:135-146)
        model.s[:] = model.sm.Einv * model.b - model.A @ x0
    elif model.is_scaled:
        xs = model.sm.Dinv * x0
        model.s[:] = model.sm.Einv * (model.b - model.A @ xs)
    else:
        model.s[:] = model.b - model.A @ x0


def warm_start_slack(model: Model, s0):
    model.s[:] = np.array(s0, dtype=np.float64)


def warm_start_dual(model: Model, y0):
    model.mu[:] = -np.array(y0, dtype=np.float64)                 # src/interface.jl:167


def update(model: Model, q=None, b=None):
    """`COSMO.update!` (src/interface.jl:187-211)."""
    if not model.is_assembled:
        raise RuntimeError("Model has to be assembled once before one can start updating q or b.")
    if q is not None:
        q = np.array(q, dtype=np.float64)
        if q.size != model.n:
            raise ValueError("The dimension of q, does not agree with the model dimension, n.")
        model.q = (model.sm.D * q) * model.sm.c if model.is_scaled else q.copy()
    if b is not None:
        b = np.array(b, dtype=np.float64)
        if b.size != model.m:
            raise ValueError("The dimension of b, does not agree with the model dimension, m.")
        model.b = model.sm.E * b if model.is_scaled else b.copy()
    if model.handle is not None:
        model.handle.update_qb(model.q if q is not None else None, model.b if b is not None else None)


def _params_from_settings(h, st: Settings):
    p = _ffi.default_params()
    kkt = st.kkt_solver
    kw = {}
    if isinstance(kkt, OptionsFactory):
        kw = kkt.kwargs
        kkt = kkt.solver
    if kkt == QdldlKKTSolver:
        raise NotImplementedError("QdldlKKTSolver is the reference's CPU direct solver (config 1, plumbing only); "
                                  "the MI355X path provides CGIndirectKKTSolver / MINRESIndirectKKTSolver")
    if kkt not in _KKT_KIND:
        raise ValueError("unknown kkt_solver %r" % (kkt,))
    p.kkt_kind = _KKT_KIND[kkt]
    p.tol_constant = float(kw.get("tol_constant", 1.0))
    p.tol_exponent = float(kw.get("tol_exponent", 1.5))
    p.sigma, p.alpha, p.rho = st.sigma, st.alpha, st.rho
    p.eps_abs, p.eps_rel = st.eps_abs, st.eps_rel
    p.eps_prim_inf, p.eps_dual_inf = st.eps_prim_inf, st.eps_dual_inf
    p.rho_min, p.rho_max, p.rho_tol = st.RHO_MIN, st.RHO_MAX, st.RHO_TOL
    p.rho_eq_over_rho_ineq = st.RHO_EQ_OVER_RHO_INEQ
    p.adaptive_rho_tolerance = st.adaptive_rho_tolerance
    p.cosmo_infty_min_scaling = st.COSMO_INFTY * st.MIN_SCALING
    p.time_limit = st.time_limit
    p.obj_true, p.obj_true_tol = st.obj_true, st.obj_true_tol
    p.max_iter = st.max_iter
    p.adaptive_rho_max_adaptions = min(st.adaptive_rho_max_adaptions, 2 ** 62)
    p.check_termination = st.check_termination
    p.check_infeasibility = st.check_infeasibility
    p.adaptive_rho = 1 if st.adaptive_rho else 0
    p.adaptive_rho_interval = st.adaptive_rho_interval
    p.adaptive_rho_fraction = st.adaptive_rho_fraction          # read only with adaptive_rho_interval == 0 (solver.jl:244-256)
    p.unscale_residuals = 1 if st.scaling != 0 else 0
    return p


def _install_accelerator(h, st: Settings):
    """`_make_accelerator!` (src/setup.jl:10-16)."""
    acc, kw = st.accelerator, {}
    if isinstance(acc, OptionsFactory):
        acc, kw = acc.solver, acc.kwargs
    if acc is None or acc is EmptyAccelerator:
        return
    if not (isinstance(acc, type) and issubclass(acc, AndersonAccelerator)):
        raise ValueError("unknown accelerator %r" % (acc,))
    act = kw.get("activation_reason", st.accelerator_activation)
    acc_kw = dict(start_accuracy=act.start_accuracy) if isinstance(act, AccuracyActivation) else dict(start_iter=int(act))
    h.set_accelerator(acc.accel_kind, mem=kw.get("mem", 15), min_mem=kw.get("min_mem", 3), safeguard=st.safeguard,
                      safeguard_tol=st.safeguard_tol, **acc_kw)


def setup(model: Model):
    """`setup!` (src/setup.jl:18-64): scaling once, then hand the scaled problem to the device library
    (the `_make_kkt_solver!` step, :1-7, is where the reference constructs its AbstractKKTSolver plugin)."""
    st = model.settings

    def make_handle():
        h = _ffi.Handle(st.device, dtype=getattr(model, "dtype", np.float64))
        h.set_problem(model.P, model.q, model.A, model.b)
        bl = np.concatenate([K.l for K in model.sets if K.kind == _ffi.BOX] or [np.zeros(0)])
        bu = np.concatenate([K.u for K in model.sets if K.kind == _ffi.BOX] or [np.zeros(0)])
        h.set_cones([K.kind for K in model.sets], [K.dim for K in model.sets], bl, bu,
                    cone_param=[getattr(K, "alpha", 0.0) for K in model.sets])
        for k, K in enumerate(model.sets):
            if K.kind == _ffi.CUSTOM:
                h.set_custom_cone(k, K.project, K.in_dual if callable(K.in_dual) else None,
                                  K.in_pol_recc if callable(K.in_pol_recc) else None)
        if st.psd_projection not in ("sign", "eigen"):
            raise ValueError("Settings.psd_projection: 'sign' or 'eigen'")
        if st.psd_projection == "eigen":
            h.set_psd_projection(_ffi.PSD_PROJECTION_EIGEN)
        return h

    on_device = (st.scaling != 0 and not model.is_scaled and model.handle is None and st.device_scaling
                 and (abs(model.P - model.P.T)).nnz == 0)
    if on_device:
        # scale_ruiz! on the device-resident problem (csrc/scaling.hip); the host keeps only D, E, c for the O(n+m)
        # pre/post-processing (scale_variables!, reverse_scaling!, update!) and the scaled q, b, Box bounds
        h = make_handle()
        D, E, c = h.scale_ruiz(st.scaling, st.MIN_SCALING, st.MAX_SCALING)
        model.sm = ScaleMatrices(D, 1.0 / D, E, 1.0 / E, c, 1.0 / c)
        model.q = (D * model.q) * c
        model.b = E * model.b
        off = 0
        for K in model.sets:
            if K.kind == _ffi.BOX:
                K.l *= E[off:off + K.dim]; K.u *= E[off:off + K.dim]
            off += K.dim
        model.is_scaled = True
        model.device_scaled = True            # host copies of P and A stay unscaled
        h.set_params(_params_from_settings(h, st))
        _install_accelerator(h, st)
        model.handle = h
    elif st.scaling != 0 and not model.is_scaled:
        model.sm = scale_ruiz(model.P, model.q, model.A, model.b, model.sets, st)
        model.is_scaled = True
    elif model.sm is None:
        model.sm = ScaleMatrices(np.ones(model.n), np.ones(model.n), np.ones(model.m), np.ones(model.m), 1.0, 1.0)
    sm = model.sm
    # scale_variables! (src/scaling.jl:118-123)
    model.x = sm.Dinv * model.x
    model.mu = (sm.Einv * model.mu) * sm.c
    model.s = sm.E * model.s
    if model.handle is None:
        h = make_handle()
        h.set_params(_params_from_settings(h, st))                 # set_rho_vec! happens inside (first solve only)
        h.set_scaling_full(sm.D, sm.Dinv, sm.E, sm.Einv, sm.c, sm.cinv)
        _install_accelerator(h, st)
        model.handle = h


def _chordal_decomposition(model: Model):
    """`chordal_decomposition!` (src/chordal_decomposition/chordal_decomposition.jl:10-38) through libcosmo_chordal.so: replaces the
    model's (P, q, A, b, sets) by the clique-tree transformed problem; `model.chordal` keeps what reverse_decomposition! needs."""
    from . import _chordal
    dk = (_ffi.PSD_TRIANGLE,) if model.settings.compact_transformation else (_ffi.PSD_TRIANGLE, _ffi.PSD_SQUARE)
    if not any(K.kind in dk and K.dim > 1 for K in model.sets):
        return
    ms, kw = model.settings.merge_strategy, {}
    if isinstance(ms, OptionsFactory):
        ms, kw = ms.solver, ms.kwargs
    dec = _chordal.Decomposition(model.A, model.b, [K.kind for K in model.sets], [K.dim for K in model.sets], merge_strategy=ms.code,
                                 t_fill=kw.get("t_fill", 8), t_size=kw.get("t_size", 8), compact=model.settings.compact_transformation)
    if dec.num_decomposed == 0:                                   # ws.ci.decompose = false (:33-35)
        dec.close()
        return
    new_sets = []
    for kind, dim, orig, clq in zip(dec.kinds, dec.dims, dec.cone_map, dec.clique_of):
        if int(orig) == 0:                                        # the ZeroSet(m) block of the traditional transformation
            new_sets.append(ZeroSet(int(dim)))
        elif int(clq) > 0:                                        # one clique of a decomposed cone
            new_sets.append(PsdConeTriangle(int(dim)) if kind == _ffi.PSD_TRIANGLE else PsdCone(int(dim)))
        else:
            new_sets.append(_copy_set(model.sets[int(orig) - 1]))
    nov = dec.n_new - model.n
    model.P = sp.block_diag([model.P, sp.csc_matrix((nov, nov))], format="csc")      # transformations.jl:193-194
    model.q = np.concatenate([model.q, np.zeros(nov)])
    model.A, model.b, model.sets = dec.A.tocsc(), dec.b.copy(), new_sets
    model.n, model.m = dec.n_new, dec.m_new
    model.x = np.zeros(model.n); model.s = np.zeros(model.m); model.mu = np.zeros(model.m)   # pre_allocate_variables! (:28)
    model.chordal = dec


def optimize(model: Model, dist=None, shard: str = "rows") -> Result:
    """`COSMO.optimize!` (src/solver.jl:78-203) with the `while` loop running on the MI355X.  With an initialised
    torch.distributed module as `dist` (one process per GPU) the problem is sharded over the ranks: shard="rows" (default; the reduced
    CG solvers) gives every rank its cones AND their rows (setup_row_sharding: one all-reduce of an n-vector per iteration),
    shard="cones" only the projections (setup_clique_sharding: one all-gather of s per iteration).  Every rank returns the same Result."""
    import time
    if not model.is_assembled:
        raise RuntimeError("The model has to be assembled! / set! before optimize!() can be called.")
    if model.settings.persistent_kernel and dist is None and model.handle is None and _fits_persistent_kernel(model):
        return _solve_shard_on_device([model], model.settings.device)[0]
    t0 = time.perf_counter()
    fresh = model.handle is None
    if fresh and model.settings.decompose and getattr(model, "chordal", None) is None:
        _chordal_decomposition(model)                             # chordal_decomposition!(ws) (src/solver.jl:88-94)
    setup(model)
    if dist is not None and dist.get_world_size() > 1 and fresh:
        kkt = model.settings.kkt_solver.solver if isinstance(model.settings.kkt_solver, OptionsFactory) else model.settings.kkt_solver
        rows_ok = kkt in (CGIndirectKKTSolver, CGSingleReductionKKTSolver, CGJacobiKKTSolver, IndirectReducedKKTSolverMINRES)
        if shard == "rows" and rows_ok:
            setup_row_sharding(model, dist)
        else:
            setup_clique_sharding(model, dist)
    t_setup = time.perf_counter() - t0
    h, sm, n = model.handle, model.sm, model.n
    h.set_iterates(model.x, model.s, model.mu)                    # solver.jl:128-129
    if model.settings.adaptive_rho and model.settings.adaptive_rho_interval == 0:
        h.set_setup_time(t_setup)                                 # ws.times.setup_time, what the automatic rho interval is measured against (solver.jl:246)
    r = h.optimize()                                              # solver.jl:137-176
    if model.settings.adaptive_rho and model.settings.adaptive_rho_interval == 0:
        model.settings.adaptive_rho_interval = h.rho_interval()[0]   # the reference writes the chosen interval into the settings (solver.jl:249-254)
    w, w_prev, s, mu = h.get_iterates()
    model.is_optimized = True
    x = w_prev[:n].copy()                                          # x is a view of w_prev (src/types.jl:274)
    if model.settings.scaling != 0:                               # reverse_scaling! (src/scaling.jl:170-179)
        x = sm.D * x
        s = sm.Einv * s
        mu = (sm.E * mu) * sm.cinv
    model.x, model.s, model.mu = x.copy(), s.copy(), mu.copy()
    dec = getattr(model, "chordal", None)
    if dec is not None:                                           # reverse_decomposition!(ws, settings) (src/solver.jl:184-190)
        x = x[:dec.n]
        s, mu = dec.reverse(s, mu, complete_dual=model.settings.complete_dual)
    info = ResultInfo(r.r_prim, r.r_dual, r.max_norm_prim, r.max_norm_dual,
                      [r.rho_updates[i] for i in range(min(r.n_rho_updates, _ffi.MAX_RHO_UPDATES))])
    times = ResultTimes(time.perf_counter() - t0, t_setup, r.iter_time, r.proj_time)
    return Result(x=x, y=-mu, s=s, obj_val=r.cost, iter=int(r.iter), status=_ffi.STATUS_NAMES[r.status], info=info,
                  times=times, kkt_iters_total=int(r.kkt_iters_total), safeguarding_iter=int(r.safeguarding_iter))


# ---------------------------------------------------------------------------------------------------------------------
# batches of independent problems (BASELINE config 3) and their sharding over the GPUs of a node
# ---------------------------------------------------------------------------------------------------------------------
def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of `n_items` independent problems for `rank` of `world` (no exchange step:
    SURVEY.md 8e, batches shard naturally)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def prepare_batch(models: Sequence[Model], device: int):
    """setup! of every problem of a shard (host Ruiz scaling per problem, as the reference does per optimize! call) and upload
    into one `_ffi.Batch` (csrc/batch.hip), iterates set.  Returns (batch, settings actually used)."""
    if not models:
        raise ValueError("prepare_batch: empty shard")
    n, m = models[0].n, models[0].m
    kinds = [K.kind for K in models[0].sets]; dims = [K.dim for K in models[0].sets]
    for md in models:
        if not md.is_assembled:
            raise RuntimeError("The model has to be assembled! / set! before optimize!() can be called.")
        if (md.n, md.m) != (n, m) or [K.kind for K in md.sets] != kinds or [K.dim for K in md.sets] != dims or \
                [getattr(K, "alpha", 0.0) for K in md.sets] != [getattr(K, "alpha", 0.0) for K in models[0].sets]:
            raise ValueError("optimize_batch: all problems of a batch must share n, m and the cone structure (kinds, dimensions, power-cone exponents)")
    st = models[0].settings
    for md in models[1:]:
        if md.settings != st:
            raise ValueError("optimize_batch: all problems of a batch share ONE Settings object (per-problem settings are not supported)")
    B = _ffi.Batch(len(models), n, m, device, dtype=getattr(models[0], "dtype", np.float64))
    _install_accelerator(B, st)                           # before set_params: the accelerated loop runs inside the persistent kernels (csrc/batch.hip)
    bl, bu = [], []
    for k, md in enumerate(models):                      # setup! per problem (scaling on the host, as in the reference)
        if st.scaling != 0 and not md.is_scaled:
            md.sm = scale_ruiz(md.P, md.q, md.A, md.b, md.sets, md.settings); md.is_scaled = True
        elif md.sm is None:
            md.sm = ScaleMatrices(np.ones(n), np.ones(n), np.ones(m), np.ones(m), 1.0, 1.0)
        md.x = md.sm.Dinv * md.x; md.mu = (md.sm.Einv * md.mu) * md.sm.c; md.s = md.sm.E * md.s
        B.set_problem(k, md.P, md.q, md.A, md.b)
        B.set_scaling(k, md.sm.Dinv, md.sm.Einv, md.sm.cinv)
        bl += [K.l for K in md.sets if K.kind == _ffi.BOX]; bu += [K.u for K in md.sets if K.kind == _ffi.BOX]
    B.set_cones(kinds, dims, np.concatenate(bl) if bl else None, np.concatenate(bu) if bu else None,
                cone_param=[getattr(K, "alpha", 0.0) for K in models[0].sets])
    B.set_params(_params_from_settings(None, st))
    B.set_iterates(np.concatenate([md.x for md in models]), np.concatenate([md.s for md in models]),
                   np.concatenate([md.mu for md in models]))
    return B, st


def _structure_key(md: Model):
    return (md.n, md.m, tuple(K.kind for K in md.sets), tuple(K.dim for K in md.sets), tuple(getattr(K, "alpha", 0.0) for K in md.sets))


def _fits_persistent_kernel(md: Model) -> bool:
    """Settings.persistent_kernel: would the LDS-resident batch kernels take this ONE model?  (A model they only take through their streaming form, or
    one whose PSD cones the front end could decompose, stays on the single-problem handle.)"""
    if not _batch_kernels_take(md) or getattr(md, "chordal", None) is not None:
        return False
    if md.settings.decompose and any(K.kind in (_ffi.PSD_SQUARE, _ffi.PSD_TRIANGLE) for K in md.sets):
        return False
    if md.n > 1024 or md.m > 2048:
        return False
    isz = 4 if getattr(md, "dtype", np.float64) == np.float32 else 8
    image = (isz + 6) * md.A.nnz + (isz + 2) * md.P.nnz + isz * (md.n + md.m) + 4 * (md.n + md.m)     # values + index entries of A, A', P; work vectors; row pointers
    return image <= 150 * 1024 and md.A.nnz <= 65535 and md.P.nnz <= 65535


def _check_batch_psd_projection(models) -> None:
    """psd_projection = "eigen" in batch mode: the persistent kernels project cones of side <= 64 by Jacobi eigensolvers anyway; a member with a larger
    cone would run on a group-internal handle, which has no such switch."""
    for md in models:
        if md.settings.psd_projection == "eigen":
            for K in md.sets:
                if K.kind in (_ffi.PSD_SQUARE, _ffi.PSD_TRIANGLE):
                    d = int(round(np.sqrt(K.dim))) if K.kind == _ffi.PSD_SQUARE else int((np.sqrt(1 + 8 * K.dim) - 1) // 2)
                    if d > 64:
                        raise NotImplementedError("optimize_batch with psd_projection='eigen' and a PSD cone of side %d > 64: solve such models with optimize()" % d)


def _batch_kernels_take(md: Model) -> bool:
    """What cosmo_hip_batch_* accepts (csrc/batch.hip): CG solver kinds, the cone types of batch mode with PSD cones of side <= 64, a fixed rho interval."""
    st = md.settings
    kkt = st.kkt_solver.solver if isinstance(st.kkt_solver, OptionsFactory) else st.kkt_solver
    if kkt not in (CGIndirectKKTSolver, CGSingleReductionKKTSolver, CGJacobiKKTSolver) or (st.adaptive_rho and st.adaptive_rho_interval == 0):
        return False
    acc = st.accelerator.solver if isinstance(st.accelerator, OptionsFactory) else st.accelerator
    if isinstance(acc, type) and issubclass(acc, AndersonAccelerator) and acc.accel_kind != _ffi.ACCEL_ANDERSON:
        return False                      # the persistent kernels carry the default Type2{QRDecomp} / RestartedMemory variant only
    for K in md.sets:
        if K.kind in (_ffi.PSD_SQUARE, _ffi.PSD_TRIANGLE):
            d = int(round(np.sqrt(K.dim))) if K.kind == _ffi.PSD_SQUARE else int((np.sqrt(1 + 8 * K.dim) - 1) // 2)
            if d > 64:
                return False
        elif K.kind not in (_ffi.ZERO, _ffi.NONNEG, _ffi.BOX, _ffi.SOC, _ffi.EXP, _ffi.DUAL_EXP, _ffi.POW, _ffi.DUAL_POW):
            return False
    return True


def prepare_batch_group(models: Sequence[Model], device: int):
    """setup! of every problem of a shard whose problems differ in structure, uploaded into one `_ffi.BatchGroup` (csrc/batch_group.hip: the library
    partitions them into classes of identical (n, m, cones) and solves the classes concurrently), iterates set.  One Settings object for all."""
    import time
    t_start = time.perf_counter()
    st = models[0].settings
    for md in models:
        if not md.is_assembled:
            raise RuntimeError("The model has to be assembled! / set! before optimize!() can be called.")
        if md.settings != st:
            raise ValueError("optimize_batch: all problems of a batch share ONE Settings object (per-problem settings are not supported)")
    G = _ffi.BatchGroup(len(models), device, dtype=getattr(models[0], "dtype", np.float64))
    _install_accelerator(G, st)
    for k, md in enumerate(models):
        n, m = md.n, md.m
        if st.scaling != 0 and not md.is_scaled:
            md.sm = scale_ruiz(md.P, md.q, md.A, md.b, md.sets, md.settings); md.is_scaled = True
        elif md.sm is None:
            md.sm = ScaleMatrices(np.ones(n), np.ones(n), np.ones(m), np.ones(m), 1.0, 1.0)
        md.x = md.sm.Dinv * md.x; md.mu = (md.sm.Einv * md.mu) * md.sm.c; md.s = md.sm.E * md.s
        G.set_problem(k, md.P, md.q, md.A, md.b)
        G.set_scaling_full(k, md.sm.D, md.sm.Dinv, md.sm.E, md.sm.Einv, md.sm.c, md.sm.cinv)      # (D, E themselves: a member on its own handle tests its certificates with them)
        bl = [K.l for K in md.sets if K.kind == _ffi.BOX]; bu = [K.u for K in md.sets if K.kind == _ffi.BOX]
        G.set_cones(k, [K.kind for K in md.sets], [K.dim for K in md.sets], np.concatenate(bl) if bl else None, np.concatenate(bu) if bu else None,
                    cone_param=[getattr(K, "alpha", 0.0) for K in md.sets])
    prm = _params_from_settings(None, st)
    if st.adaptive_rho and st.adaptive_rho_interval == 0:
        prm.setup_time = time.perf_counter() - t_start        # ws.times.setup_time of the problems that run on their own handle (solver.jl:246)
    G.set_params(prm)
    for k, md in enumerate(models):
        G.set_iterates(k, md.x, md.s, md.mu)
    return G, st


LAST_BATCH_INFO: dict = {}     # diagnostics of the last _solve_shard_on_device call (tests / timing labs)


def _solve_shard_on_device(models: Sequence[Model], device: int) -> List[Result]:
    """All problems of the shard concurrently on one MI355X (one persistent workgroup per problem, csrc/batch.hip).  Problems of ONE structure take
    the batch directly; a mixed list goes through the batch group (one batch per structure class, all classes concurrently)."""
    import time
    if not models:
        return []
    _check_batch_psd_projection(models)
    t0 = time.perf_counter()
    # one structure the persistent kernels take -> the batch directly; anything else (several structures, a PSD cone of side > 64, a MINRES solver kind,
    # the automatic rho interval) -> the group, which gives every structure class its batch or, where the batch kernels refuse, one handle per problem
    mixed = len({_structure_key(md) for md in models}) > 1 or not _batch_kernels_take(models[0])
    B, st = prepare_batch_group(models, device) if mixed else prepare_batch(models, device)
    t_setup = time.perf_counter() - t0
    t1 = time.perf_counter()
    rs = B.optimize()
    LAST_BATCH_INFO.clear()
    LAST_BATCH_INFO.update(dict(problems=len(models), mixed=bool(mixed), setup_seconds=t_setup, optimize_seconds=time.perf_counter() - t1))
    if mixed:
        LAST_BATCH_INFO.update(B.run_info())                       # worker threads / jobs of the group's bounded pool, structure classes
        LAST_BATCH_INFO["own_handle_members"] = int(np.sum(B.class_info(with_modes=True)[2] == 1))     # members the batch kernels refused
    out = []
    for k, (md, r) in enumerate(zip(models, rs)):
        w, w_prev, s, mu = B.get_iterates(k)
        x = w_prev[:md.n].copy()
        if st.scaling != 0:
            x = md.sm.D * x; s = md.sm.Einv * s; mu = (md.sm.E * mu) * md.sm.cinv
        md.x, md.s, md.mu = x.copy(), s.copy(), mu.copy()
        md.is_optimized = True
        if mixed and md.settings.adaptive_rho and md.settings.adaptive_rho_interval == 0:
            chosen = B.rho_interval(k)[0]                          # the reference writes the chosen interval into the settings (src/solver.jl:249-254)
            if chosen > 0:
                md.settings.adaptive_rho_interval = chosen
        info = ResultInfo(r.r_prim, r.r_dual, r.max_norm_prim, r.max_norm_dual,
                          [r.rho_updates[i] for i in range(min(r.n_rho_updates, _ffi.MAX_RHO_UPDATES))])
        out.append(Result(x=x, y=-mu, s=s, obj_val=r.cost, iter=int(r.iter), status=_ffi.STATUS_NAMES[r.status], info=info,
                          times=ResultTimes(time.perf_counter() - t0, t_setup, r.iter_time, 0.0), kkt_iters_total=int(r.kkt_iters_total),
                          safeguarding_iter=int(r.safeguarding_iter)))
    B.close()
    return out


def optimize_batch(models: Sequence[Model], device: Optional[int] = None, dist=None, solve_shard=None) -> List[Result]:
    """`optimize!` for a batch of independent models.  Under torch.distributed (`dist` = the initialised module) every rank
    solves its contiguous shard on its own GPU and the per-problem results are exchanged once at the end
    (all_gather_object); there is no collective inside the loop.  `solve_shard(models, device)` is injectable for tests."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    lo, hi = shard_range(len(models), rank, world)
    if device is not None:
        dev = device
    elif dist is not None:                                        # one process per GPU of ONE node: the local rank, not the global one
        import os
        dev = int(os.environ.get("LOCAL_RANK", rank))
    else:
        dev = 0
    fn = solve_shard or _solve_shard_on_device
    local = fn(list(models[lo:hi]), dev)
    if dist is None or world == 1:
        return local
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    out: List[Result] = []
    for part in gathered:
        out.extend(part)
    return out


def balance_cones(dims: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time partition of PSD cliques over ranks by projection cost ~ d^3 (option 1:
    replicated affine step, sharded projection).  Returns the cone indices owned by every rank (deterministic)."""
    order = sorted(range(len(dims)), key=lambda i: (-int(dims[i]) ** 3, i))
    load = [0] * world
    owner: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[r].append(i)
        load[r] += int(dims[i]) ** 3
    return [sorted(o) for o in owner]


def cone_costs(sets: Sequence[AbstractConvexSet]) -> List[int]:
    """Projection cost model per cone of the CONE-sharded run: ~d^3 for a PSD cone of side d, ~dim for a second-order cone, 0 for the cones
    that every rank projects (elementwise kernel; exponential / power / user cones, which csrc/cone3.hip / custom.hip do not shard)."""
    out = []
    for K in sets:
        if K.kind in (_ffi.PSD_SQUARE, _ffi.PSD_TRIANGLE) and K.dim > 1:
            out.append(int(K.sqrt_dim) ** 3)
        elif K.kind == _ffi.PSD_TRIANGLE_COMPLEX and K.dim > 1:
            out.append((2 * int(K.sqrt_dim)) ** 3)               # projected through its real 2r x 2r embedding
        elif K.kind == _ffi.SOC:
            out.append(int(K.dim))
        else:
            out.append(0)
    return out


def row_shard_costs(sets: Sequence[AbstractConvexSet]) -> List[int]:
    """Cost model of the row-sharded run: the projection cost of cone_costs() plus the cone's rows (the row-local kernels -- rhs, A x_tl,
    s_tl, w_s, primal residual -- stream ~100 B per row; a d^3 flop projection at ~1/3 of the matrix peak makes one row worth ~8 'd^3 units')."""
    newton = (_ffi.EXP, _ffi.DUAL_EXP, _ffi.POW, _ffi.DUAL_POW)          # a Newton / bisection solve per 3-vector; row-sharded runs do shard them
    return [c + 8 * int(K.dim) + (64 if K.kind in newton else 0) for c, K in zip(cone_costs(sets), sets)]


def partition_cones_contiguous(costs: Sequence[int], world: int) -> List[int]:
    """Boundaries first_cone[0..world] of the contiguous partition of the cone list that minimises the maximum load
    (binary search on the bottleneck + greedy feasibility).  Contiguous ranges keep every rank's rows one slice of `s`, so
    the exchange step is one in-place broadcast per rank."""
    n = len(costs)
    if world <= 1 or n == 0:
        return [0] + [n] * max(world, 1)
    if sum(costs) == 0:                                          # nothing to balance: an even split of the cone list
        return [(n * r) // world for r in range(world)] + [n]
    lo, hi = max(costs) if costs else 0, sum(costs)

    def cuts(limit):
        b, acc = [0], 0
        for i, c in enumerate(costs):
            if acc + c > limit and acc > 0:
                b.append(i); acc = 0
            acc += c
        return b
    while lo < hi:
        mid = (lo + hi) // 2
        if len(cuts(mid)) <= world:
            hi = mid
        else:
            lo = mid + 1
    b = cuts(lo)
    b += [n] * (world + 1 - len(b))
    return b


def setup_clique_sharding(model: Model, dist) -> None:
    """One communicator per handle (one process per GPU): rank 0 creates the RCCL unique id, the host layer distributes it,
    every rank takes a contiguous cone range balanced by cone_costs()."""
    _comm_for(model, dist)
    model.handle.set_cone_shard(partition_cones_contiguous(cone_costs(model.sets), dist.get_world_size()))


def _comm_for(model: Model, dist) -> None:
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = [_ffi.Handle.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    model.handle.comm_init(rank, world, uid[0])


def setup_row_sharding(model: Model, dist) -> None:
    """Row sharding (csrc/rowshard.hip): a contiguous cone range per rank balanced by row_shard_costs(); the handle keeps its cones, their
    rows of A / columns of A' and the slices of every row vector; the n-side (CG included) is replicated."""
    _comm_for(model, dist)
    model.handle.set_row_shard(partition_cones_contiguous(row_shard_costs(model.sets), dist.get_world_size()))
