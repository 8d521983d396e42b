/* cosmo_hip.h -- C ABI of libcosmo_hip.so: the MI355X (gfx950) implementation of COSMO's per-iteration
 * ADMM hot path.  This header is the drop-in boundary.
 *
 * The reference (oxfordcontrol/COSMO.jl v0.8.11, pure Julia) has no FFI for this path; its plugin
 * surfaces are Julia multiple-dispatch interfaces.  Each entry point below names the reference
 * interface (file:line under /root/reference) that a Julia `ccall` binding replaces with it; the
 * binding itself is shown in INTEGRATION.md and julia/CosmoHIP.jl.
 *
 * Conventions
 *  - every function returns an int32 status (COSMO_HIP_OK == 0); nothing throws, nothing calls back;
 *  - host arrays are caller-owned and never retained after return; device memory is owned by the handle;
 *  - sparse matrices are passed exactly as Julia stores SparseMatrixCSC{Float64,Int64}: CSC, 1-based
 *    colptr[ncols+1], rowval[nnz], nzval[nnz], row indices sorted within a column;
 *  - the problem is in COSMO's INTERNAL, already scaled form  min 1/2 x'Px + q'x  s.t.  A x + s = b, s in K
 *    (src/interface.jl:478-484, src/scaling.jl:21-116);
 *  - all vectors are dense Float64; n = #variables, m = #constraint rows;
 *  - one handle is driven by one host thread at a time; several handles may be driven concurrently.
 */
#ifndef COSMO_HIP_H
#define COSMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element type of every DATA array that crosses this ABI (matrix values, vectors, iterates, bounds).  libcosmo_hip.so is built
 * with double (COSMO.Model{Float64}); libcosmo_hip_f32.so is the SAME source and the SAME symbol names built with
 * -DCOSMO_HIP_REAL_FLOAT (COSMO.Model{Float32}, src/types.jl:348; the reference's test-suite runs every test for both types,
 * test/run_tests.jl).  Scalars (settings, residuals, times) stay double in both libraries: a Float32 setting converts exactly. */
#ifdef COSMO_HIP_REAL_FLOAT
typedef float cosmo_hip_real;
#else
typedef double cosmo_hip_real;
#endif

typedef struct cosmo_hip_handle cosmo_hip_handle;
/* The libraries are built with -fvisibility=hidden: the functions tagged COSMO_HIP_API below are their ONLY exported symbols
 * (tests/test_abi_and_host.py compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#define COSMO_HIP_API __attribute__((visibility("default")))
#else
#define COSMO_HIP_API
#endif

/* ---- status codes -------------------------------------------------------------------------------- */
enum {
  COSMO_HIP_OK = 0,
  COSMO_HIP_ERR_INVALID = 1,     /* bad argument / call order                                          */
  COSMO_HIP_ERR_HIP = 2,         /* a HIP runtime call failed (no device, out of memory, ...)           */
  /* 3 and 4 are retired (they named a CG breakdown and a non-finite iterate; the reference raises neither on this path: its cg! does
     not test u'Lu, and a NaN iterate makes norm(., Inf) NaN => has_converged false => Max_iter_reached, src/residuals.jl:30-53,
     98-140 -- which is what the device loop reports, tests/test_gpu_nan_residuals.py) */
  COSMO_HIP_ERR_EIG = 5,         /* eigen-solver did not converge (reference: chklapackerror,
                                    src/convexset.jl:186)                                              */
  COSMO_HIP_ERR_UNSUPPORTED = 6, /* feature not built (e.g. a cone type outside SURVEY 8a)              */
  COSMO_HIP_ERR_COMM = 7         /* RCCL failure                                                        */
};

/* ---- cone types: AbstractConvexSet subtypes on the hot path (src/convexset.jl) -------------------- */
enum {
  COSMO_HIP_ZERO = 0,        /* ZeroSet          src/convexset.jl:16-28   */
  COSMO_HIP_NONNEG = 1,      /* Nonnegatives     src/convexset.jl:52-74   */
  COSMO_HIP_BOX = 2,         /* Box              src/convexset.jl:803-847 */
  COSMO_HIP_SOC = 3,         /* SecondOrderCone  src/convexset.jl:92-114  */
  COSMO_HIP_PSD_SQUARE = 4,  /* PsdCone          src/convexset.jl:271-321 */
  COSMO_HIP_PSD_TRIANGLE = 5,/* PsdConeTriangle  src/convexset.jl:362-412 */
  COSMO_HIP_EXP = 6,         /* ExponentialCone      src/convexset.jl:497-605  (dim 3, MAX_ITERS 100, EXP_TOL 1e-8) */
  COSMO_HIP_DUAL_EXP = 7,    /* DualExponentialCone  src/convexset.jl:735-779  (Moreau decomposition)               */
  COSMO_HIP_POW = 8,         /* PowerCone(alpha)     src/convexset.jl:607-726  (dim 3, MAX_ITERS 20, POW_TOL 1e-8)  */
  COSMO_HIP_DUAL_POW = 9,    /* DualPowerCone(alpha) src/convexset.jl:748-779                                       */
  COSMO_HIP_PSD_TRIANGLE_COMPLEX = 10, /* PsdConeTriangle{T, Complex{T}}(r^2): Hermitian r x r matrices, vector = svec of the
                                real part (r(r+1)/2 entries) followed by sqrt(2) * imaginary parts of the strict upper
                                triangle, column by column (src/convexset.jl:345-380, 444-490)                          */
  COSMO_HIP_CUSTOM = 11      /* a user subtype of AbstractConvexCone{T} (src/projections.jl:4-5): its project! runs on the
                                host through cosmo_hip_set_custom_cone (docs/src/literate/custom_cone.jl:9-17)          */
};

/* ---- KKT solver kinds: AbstractKKTSolver subtypes (src/linear_solver/kktsolver_indirect.jl) -------- */
enum {
  COSMO_HIP_KKT_CG = 0,             /* CGIndirectKKTSolver      :173-178 (IndirectReducedKKTSolver, :CG): the literal recurrence of
                                       IterativeSolvers v0.9's cg! everywhere (cosmo_hip_kkt_recurrence names the kernels that run it)  */
  COSMO_HIP_KKT_MINRES_REDUCED = 1, /* IndirectReducedKKTSolver :3-88 with solver_type = :MINRES            */
  COSMO_HIP_KKT_MINRES = 2,         /* MINRESIndirectKKTSolver  :180-185 (IndirectKKTSolver, full KKT)      */
  COSMO_HIP_KKT_CG_SR = 3           /* OPT-IN, no reference counterpart: the reduced CG solve as single-reduction (Chronopoulos-Gear) CG --
                                       same operator, stopping rule and warm start as COSMO_HIP_KKT_CG, algebraically equal iterates, two
                                       launches per Krylov iteration (ONE on an assembled operator); not bit-comparable with the literal
                                       recurrence (csrc/cg_sr.hip).  Measured in round 6 as a candidate DEFAULT for assembled operators and
                                       rejected: 12.4 vs 11.5 us per Krylov iteration on BASELINE config 5 (24-byte gathers), +1.2 % Krylov
                                       iterations at a 1e-10 stopping threshold */
  ,
  COSMO_HIP_KKT_CG_JACOBI = 4       /* OPT-IN, no reference counterpart (the reference calls cg! without a preconditioner,
                                       src/linear_solver/kktsolver_indirect.jl:70): IterativeSolvers' preconditioned recurrence (PCGIterable) with
                                       Pl = Diagonal(diag(P + sigma I + A' rho A)) on the ASSEMBLED reduced operator; same operator, warm start
                                       and true-residual stopping rule ||r||_2 <= tol_k / ||rhs||, DIFFERENT iterates (every solve ends at another
                                       point inside the same tolerance).  cosmo_hip_set_params fails with UNSUPPORTED where the operator cannot be
                                       assembled (csrc/cg_fold.hip).  Never the parity path. */
};

/* ---- solver status (Result.status symbols, src/solver.jl:113,175,312,318,338,344,353) -------------- */
enum {
  COSMO_HIP_UNDETERMINED = 0,
  COSMO_HIP_SOLVED = 1,
  COSMO_HIP_MAX_ITER_REACHED = 2,
  COSMO_HIP_UNSOLVED = 3,
  COSMO_HIP_PRIMAL_INFEASIBLE = 4,
  COSMO_HIP_DUAL_INFEASIBLE = 5,
  COSMO_HIP_TIME_LIMIT_REACHED = 6
};

/* ---- which matrix for cosmo_hip_spmv --------------------------------------------------------------- */
enum { COSMO_HIP_MAT_A = 0, COSMO_HIP_MAT_AT = 1, COSMO_HIP_MAT_P = 2 };

/* Numeric fields of COSMO.Settings that the hot path reads (src/settings.jl:101-139). */
typedef struct cosmo_hip_params {
  double sigma;                      /* 1e-6  */
  double alpha;                      /* 1.6   */
  double rho;                        /* 0.1   */
  double eps_abs, eps_rel;           /* 1e-5  */
  double eps_prim_inf, eps_dual_inf; /* 1e-4  */
  double tol_constant, tol_exponent; /* 1.0, 1.5  (kktsolver_indirect.jl:21) */
  double rho_min, rho_max;           /* RHO_MIN 1e-6, RHO_MAX 1e6 */
  double rho_tol;                    /* RHO_TOL 1e-4 */
  double rho_eq_over_rho_ineq;       /* 1e3 */
  double adaptive_rho_tolerance;     /* 5 */
  double cosmo_infty_min_scaling;    /* COSMO_INFTY * MIN_SCALING = 1e20*1e-4 (src/setup.jl:79-81) */
  double time_limit;                 /* 0 = none */
  int64_t max_iter;                  /* 5000 */
  int64_t adaptive_rho_max_adaptions;/* typemax(Int) */
  int32_t kkt_kind;                  /* COSMO_HIP_KKT_* */
  int32_t check_termination;         /* 25 */
  int32_t check_infeasibility;       /* 40 */
  int32_t adaptive_rho;              /* 1 */
  int32_t adaptive_rho_interval;     /* 40.  0 = the reference's automatic interval (src/solver.jl:244-256): once the loop has run for
                                        adaptive_rho_fraction * setup_time seconds the interval is fixed, ONCE, to the current iteration
                                        count rounded to a multiple of check_termination (at least one); from then on the device schedule
                                        is that of a fixed interval.  The host applies the rule where it knows the device's progress: at the
                                        termination checks (every iteration in accelerated runs).  Not in batch mode. */
  int32_t unscale_residuals;         /* 1 iff settings.scaling != 0 (src/residuals.jl:43) */
  double obj_true;                   /* NaN  (settings.obj_true): if set, has_converged additionally requires            */
  double obj_true_tol;               /* 1e-3 |obj_true - cost| <= obj_true_tol (src/residuals.jl:131-139)                 */
  double adaptive_rho_fraction;      /* 0.4  (settings.adaptive_rho_fraction) -- read only with adaptive_rho_interval == 0                    */
  double setup_time;                 /* 0    ws.times.setup_time of the caller's setup! in seconds (src/solver.jl:246) -- ditto               */
} cosmo_hip_params;

/* Accelerator of the fixed-point iteration (settings.accelerator, safeguard, safeguard_tol: src/settings.jl:96-98,136-138;
 * activation: src/accelerator_interface.jl:5-47).  ANDERSON = AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory,
 * NoRegularizer} of COSMOAccelerators.jl, the reference's default. */
/* Round 6: the non-default variants the reference documents (docs/src/acceleration.md:23-26, src/printing.jl:83-97) -- broyden type Type1 or
 * Type2{NormalEquations} (the mem x mem system M eta = L' f with M = L' F, L = X resp. F, solved by LU with partial pivoting) with RestartedMemory
 * or RollingMemory (the oldest column is overwritten).  Single-problem handles, also row-sharded; the persistent batch kernels carry the default
 * variant only (cosmo_hip_batch_set_accelerator returns UNSUPPORTED for the others: a group runs such members on their own handles). */
enum { COSMO_HIP_ACCEL_EMPTY = 0, COSMO_HIP_ACCEL_ANDERSON = 1,
       COSMO_HIP_ACCEL_ANDERSON_TYPE1_RESTARTED = 2, COSMO_HIP_ACCEL_ANDERSON_TYPE1_ROLLING = 3,
       COSMO_HIP_ACCEL_ANDERSON_TYPE2NE_RESTARTED = 4, COSMO_HIP_ACCEL_ANDERSON_TYPE2NE_ROLLING = 5 };
typedef struct cosmo_hip_accel_params {
  int32_t kind;          /* COSMO_HIP_ACCEL_*                                                   */
  int32_t mem;           /* 15  (with_options(..., mem = 15), src/settings.jl:136); <= 32        */
  int32_t min_mem;       /* 3   columns needed before a step is attempted                        */
  int32_t safeguard;     /* 1   (settings.safeguard)                                             */
  int64_t start_iter;    /* 2 = ImmediateActivation; k = IterActivation(k)                       */
  double safeguard_tol;  /* 2.0 (settings.safeguard_tol)                                         */
  double eta_max;        /* 1e4: a least-squares solution with a larger 2-norm is rejected       */
  double start_accuracy; /* < 0 (default -1): unused.  >= 0: AccuracyActivation(start_accuracy) -- the accelerator is
                            switched on at the first termination check whose residuals satisfy eps_abs = eps_rel =
                            start_accuracy (src/accelerator_interface.jl:14-21,38-46); start_iter is then ignored */
} cosmo_hip_accel_params;

/* What `optimize!` returns to its caller besides the iterates (Result / ResultInfo, src/types.jl:65-112). */
#define COSMO_HIP_MAX_RHO_UPDATES 64
typedef struct cosmo_hip_result {
  int32_t status;                    /* COSMO_HIP_SOLVED ... */
  int32_t n_rho_updates;             /* length(ws.rho_updates) (>= 1) */
  int64_t iter;                      /* ADMM iterations performed: Result.iter = iter + safeguarding_iter of src/solver.jl:196 */
  int64_t kkt_iters_total;           /* sum of CG/MINRES iterations over all solves */
  int64_t kkt_solves;                /* number of solve! calls (IndirectReducedKKTSolver.iteration_counter - 1) */
  double cost;                       /* cinv * (1/2 x'Px + q'x) at the last check (src/residuals.jl:143-147) */
  double r_prim, r_dual, max_norm_prim, max_norm_dual; /* ResultInfo */
  double rho;                        /* ws.rho at exit */
  double iter_time;                  /* seconds in the while loop: ws.times.iter_time (src/solver.jl:134,169) */
  double proj_time;                  /* seconds in admm_z! projections (ws.times.proj_time) -- measured with HIP events */
  double rho_updates[COSMO_HIP_MAX_RHO_UPDATES];
  int64_t safeguarding_iter;         /* of them: extra ADMM steps after a declined accelerated candidate (Result.safeguarding_iter, src/solver.jl:201) */
} cosmo_hip_result;

/* ---- lifecycle -------------------------------------------------------------------------------------- */
/* Creates a handle bound to HIP device `device_id` (one process per GPU; the handle owns one stream). */
COSMO_HIP_API int32_t cosmo_hip_create(cosmo_hip_handle** h, int32_t device_id);
/* Idempotent.  Replaces AbstractKKTSolver free_memory! (src/linear_solver/kktsolver.jl:351; called from
 * src/solver.jl:200,206-208) and is what the Julia finalizer of the wrapper calls. */
COSMO_HIP_API int32_t cosmo_hip_destroy(cosmo_hip_handle* h);
/* Last error text of this handle (valid until the next call on it); never NULL. */
COSMO_HIP_API const char* cosmo_hip_last_error(const cosmo_hip_handle* h);
/* ABI version of the library (major*1000 + minor).  COSMO_HIP_ABI_VERSION is the version THIS header describes; the bindings generated from
 * it (cosmo.jl_amd/_abi_structs.py, julia/abi_structs.jl) carry the same number and refuse a library that reports another one: a stale
 * .so paired with newer struct mirrors would read garbage, a newer .so would write past an older caller's cosmo_hip_result. */
#define COSMO_HIP_ABI_VERSION 1004
COSMO_HIP_API int32_t cosmo_hip_version(void);
COSMO_HIP_API void cosmo_hip_default_params(cosmo_hip_params* p);

/* ---- problem data ----------------------------------------------------------------------------------- */
/* Replaces the AbstractKKTSolver constructor T(P, A, sigma, rho) (src/linear_solver/kktsolver.jl:5-11;
 * called from _make_kkt_solver!, src/setup.jl:1-7) together with the (q, b) the loop reads from ws.p
 * (src/solver.jl:137,154).  P (n x n, full symmetric storage) and A (m x n) are the SCALED matrices. */
COSMO_HIP_API int32_t cosmo_hip_set_problem(cosmo_hip_handle* h, int64_t n, int64_t m,
                              const int64_t* P_colptr, const int64_t* P_rowval, const cosmo_hip_real* P_nzval,
                              const int64_t* A_colptr, const int64_t* A_rowval, const cosmo_hip_real* A_nzval,
                              const cosmo_hip_real* q, const cosmo_hip_real* b);
/* Replaces ws.p.C::CompositeConvexSet (src/projections.jl:20-31) + get_set_indices
 * (src/convexset.jl:985-993) + classify_constraints! (src/setup.jl:75-85).  `type[k]`, `dim[k]` per cone in
 * row order; box_l/box_u are the concatenated (already E-scaled, src/convexset.jl:863-867) bounds of all Box
 * cones in order (may be NULL when there is no Box). */
COSMO_HIP_API int32_t cosmo_hip_set_cones(cosmo_hip_handle* h, int64_t ncones, const int32_t* type, const int64_t* dim,
                            const cosmo_hip_real* box_l, const cosmo_hip_real* box_u);
/* Same, plus one parameter per cone: cone_param[k] = alpha for PowerCone / DualPowerCone (0 < alpha < 1, the reference
 * throws a DomainError otherwise, src/convexset.jl:614,758), ignored for every other type.  May be NULL when the
 * composite set holds no power cone. */
COSMO_HIP_API int32_t cosmo_hip_set_cones_ex(cosmo_hip_handle* h, int64_t ncones, const int32_t* type, const int64_t* dim,
                               const cosmo_hip_real* box_l, const cosmo_hip_real* box_u, const cosmo_hip_real* cone_param);
/* ---- user-defined cones: the AbstractConvexSet plugin surface (src/projections.jl:4-5, docs/src/literate/custom_cone.jl) ----
 * project!(x, C)            -> cosmo_hip_project_fn: x is the cone's contiguous slice (dim doubles, host memory), projected in place
 * in_dual(x, C, tol) / in_pol_recc(x, C, tol) -> cosmo_hip_cone_test_fn: non-zero = member.  Optional (NULL): the cone then never
 *   certifies infeasibility ("infeasibility detection is disabled", custom_cone.jl:52-54).
 * The callbacks are invoked on the CALLING thread, from inside cosmo_hip_project / _admm_iterate / _optimize: the slice is
 * copied device -> pinned host, projected, copied back, once per iteration (the reference calls project! at the same point,
 * src/convexset.jl:885-891).  They must not call back into this library.  Custom cones are scaled by one scalar per cone
 * (rectify_scaling! fall-back, src/convexset.jl:953-954) and get the inequality rho class.  cone = 0-based index into the
 * table given to cosmo_hip_set_cones[_ex], whose type[cone] must be COSMO_HIP_CUSTOM; call after set_cones. */
typedef void (*cosmo_hip_project_fn)(cosmo_hip_real* x, int64_t dim, void* user);
typedef int32_t (*cosmo_hip_cone_test_fn)(const cosmo_hip_real* x, int64_t dim, double tol, void* user);
COSMO_HIP_API int32_t cosmo_hip_set_custom_cone(cosmo_hip_handle* h, int64_t cone, cosmo_hip_project_fn project, cosmo_hip_cone_test_fn in_dual,
                                  cosmo_hip_cone_test_fn in_pol_recc, void* user);
/* Settings fields (src/settings.jl) + initial rho vector: set_rho_vec! (src/parameters.jl:3-13).
 * rho_vec may be NULL: then it is built from p->rho and the row classes exactly as the reference does. */
COSMO_HIP_API int32_t cosmo_hip_set_params(cosmo_hip_handle* h, const cosmo_hip_params* p, const cosmo_hip_real* rho_vec);
/* Replaces update_rho!(kkt_solver, rho_vec) (src/linear_solver/kktsolver_indirect.jl:164-166; called from
 * update_rho_vec!, src/parameters.jl:85-89). */
COSMO_HIP_API int32_t cosmo_hip_update_rho(cosmo_hip_handle* h, const cosmo_hip_real* rho_vec);
/* ScaleMatrices Dinv (n), Einv (m), cinv used ONLY to unscale residuals (src/residuals.jl:43-49,66-92).
 * NULL pointers mean identity. */
COSMO_HIP_API int32_t cosmo_hip_set_scaling(cosmo_hip_handle* h, const cosmo_hip_real* Dinv, const cosmo_hip_real* Einv, double cinv);
/* Same plus D (n), E (m), c themselves, which the infeasibility certificates scale with (src/infeasibility.jl:5,35,39);
 * cosmo_hip_set_scaling derives them as reciprocals.  NULL = identity. */
COSMO_HIP_API int32_t cosmo_hip_set_scaling_full(cosmo_hip_handle* h, const cosmo_hip_real* D, const cosmo_hip_real* Dinv, const cosmo_hip_real* E, const cosmo_hip_real* Einv,
                                   double c, double cinv);
/* Replaces COSMO.update!(model; q, b) on already-scaled vectors (src/interface.jl:187-211). NULL = keep. */
/* Replaces _make_accelerator! (src/setup.jl:10-16): installs (or with kind EMPTY / NULL removes) the accelerator used by
 * cosmo_hip_optimize.  With an accelerator the loop follows src/solver.jl:140-165 including acceleration_pre! /
 * acceleration_post! (safeguarding re-does the ADMM step from the last non-accelerated point and counts it in
 * safeguarding_iter), deferred rho updates and deferred infeasibility checks (update_suggested, src/solver.jl:284-292).
 * Call after cosmo_hip_set_problem; every cosmo_hip_set_iterates restarts it (src/setup.jl:47-49). */
COSMO_HIP_API void cosmo_hip_default_accel_params(cosmo_hip_accel_params* p);
COSMO_HIP_API int32_t cosmo_hip_set_accelerator(cosmo_hip_handle* h, const cosmo_hip_accel_params* p);
/* out = {accelerated steps, safeguard accepted, safeguard declined, memory restarts, active, safeguarding_iter} */
COSMO_HIP_API int32_t cosmo_hip_get_accel_stats(cosmo_hip_handle* h, int64_t out[6]);
/* out = {restarts of the accelerator because its memory was full (RestartedMemory), restarts because rho was adapted (CA.restart!, src/solver.jl:272-275)}
 * of the last optimize; the second equals num_rho_adaptions in test/UnitTests/AccelerationTests/adaptive_rho_acc_restarts.jl:24 */
COSMO_HIP_API int32_t cosmo_hip_get_accel_restarts(cosmo_hip_handle* h, int64_t out[2]);
/* Replaces scale_ruiz! (src/scaling.jl:21-116) for callers that hand over the UNSCALED problem: call after
 * cosmo_hip_set_problem + cosmo_hip_set_cones (unscaled data and Box bounds) and before cosmo_hip_set_params.  Runs
 * `iterations` (settings.scaling) steps of the modified Ruiz equilibration on the device-resident P, A, q, b, rectifies the
 * scalings of the scalar-scaled cones (src/scaling.jl:129-142, src/convexset.jl:953-982), scales the Box bounds
 * (src/convexset.jl:863-867) and re-runs classify_constraints! on the scaled data.  The scaling matrices stay on the device
 * for the residual / infeasibility tests (as after cosmo_hip_set_scaling_full); D_out[n], E_out[m], c_out (each may be
 * NULL) return them for the caller's reverse_scaling! (src/scaling.jl:170-179).  P must be symmetric. */
COSMO_HIP_API int32_t cosmo_hip_scale_ruiz(cosmo_hip_handle* h, int64_t iterations, double min_scaling, double max_scaling, cosmo_hip_real* D_out,
                             cosmo_hip_real* E_out, double* c_out);
COSMO_HIP_API int32_t cosmo_hip_update_qb(cosmo_hip_handle* h, const cosmo_hip_real* q, const cosmo_hip_real* b);
/* Per-row rho class computed by the library: 0 = rho, 1 = rho*RHO_EQ_OVER_RHO_INEQ, 2 = RHO_MIN
 * (apply_constraint_rho_scaling!, src/parameters.jl:17-49) -- integer bookkeeping, compared bit-exactly. */
COSMO_HIP_API int32_t cosmo_hip_get_rho_classes(cosmo_hip_handle* h, int32_t* cls /* m */);
COSMO_HIP_API int32_t cosmo_hip_get_rho_vec(cosmo_hip_handle* h, cosmo_hip_real* rho_vec /* m */);

/* ---- fine-grained plugin entry points (host pointers, synchronous) ----------------------------------- */
/* Replaces solve!(kkt_solver, lhs, rhs) (src/linear_solver/kktsolver.jl:5-11, kktsolver_indirect.jl:36-88,
 * 123-162; called from admm_x!, src/solver.jl:52).  lhs, rhs have length n+m.  kkt_iters_out may be NULL. */
COSMO_HIP_API int32_t cosmo_hip_kkt_solve(cosmo_hip_handle* h, cosmo_hip_real* lhs, const cosmo_hip_real* rhs, int64_t* kkt_iters_out);
/* Replaces project!(s::SplitVector, C::CompositeConvexSet) (src/convexset.jl:885-891) on a host vector of
 * length m, in place.  psd_rank_out[k] (per cone, -1 for non-PSD cones) = nnz_lambda of rank_k_update!
 * (src/convexset.jl:247-256); soc_branch_out[k] (per cone, -1 for non-SOC) = 0 keep / 1 zero / 2 scale
 * (src/convexset.jl:104-112); for the exponential / power cones it reports the case 1..4 of their project! (in cone /
 * polar => 0 / boundary shortcut / root finding, src/convexset.jl:510-537, 626-655).  Either may be NULL. */
COSMO_HIP_API int32_t cosmo_hip_project(cosmo_hip_handle* h, cosmo_hip_real* s, int64_t* psd_rank_out, int32_t* soc_branch_out);
/* Replaces mul!(y, A, x), mul!(y, A', x), mul!(y, P, x) (src/residuals.jl:4,12,15). */
COSMO_HIP_API int32_t cosmo_hip_spmv(cosmo_hip_handle* h, int32_t which, cosmo_hip_real* y, const cosmo_hip_real* x);

/* ---- coarse device-resident loop (the performance path; replaces the body of optimize!) ------------- */
/* Warm start: w[1:n] = x0 ; w[n+1:] = 1/rho .* mu0 + s0 ; s = s0 (src/solver.jl:128-129).  NULL = zeros. */
COSMO_HIP_API int32_t cosmo_hip_set_iterates(cosmo_hip_handle* h, const cosmo_hip_real* x0, const cosmo_hip_real* s0, const cosmo_hip_real* mu0);
/* admm_x! ; admm_w! once (src/solver.jl:137-138). */
COSMO_HIP_API int32_t cosmo_hip_admm_init(cosmo_hip_handle* h);
/* n_iters times the loop body admm_z! / apply_rho_adaptation_rules! / admm_x! / admm_w!
 * (src/solver.jl:151-155) with NO termination checks; iteration numbering continues from the handle's
 * counter (so rho adaptation fires at the same iterations as in the reference). */
COSMO_HIP_API int32_t cosmo_hip_admm_iterate(cosmo_hip_handle* h, int64_t n_iters);
/* Same, but WITH check_termination! at the reference's schedule (iter % check_termination == 0 || iter == 1,
 * src/solver.jl:306) counted on the handle's absolute iteration counter; stops early when a status is decided.
 * This is the timed region of the BASELINE metric (iter_time includes the checks, src/solver.jl:134,169).
 * status_out receives COSMO_HIP_UNDETERMINED or the decided status. */
COSMO_HIP_API int32_t cosmo_hip_admm_iterate_checked(cosmo_hip_handle* h, int64_t n_iters, int32_t* status_out);
/* recover_mu! + calculate_result_info! + calculate_cost! (src/solver.jl:307-310, src/residuals.jl:30-96,
 * 143-153) on the current iterates: out = {r_prim, r_dual, max_norm_prim, max_norm_dual, cost}. */
COSMO_HIP_API int32_t cosmo_hip_residuals(cosmo_hip_handle* h, double out[5]);
/* The whole `while` loop of optimize! (src/solver.jl:137-176): init step, iterations with
 * check_termination!/adaptive rho at the reference's schedule, final recover_mu!.  Iterates stay on the
 * device; fetch them with cosmo_hip_get_iterates. */
COSMO_HIP_API int32_t cosmo_hip_optimize(cosmo_hip_handle* h, cosmo_hip_result* result);
/* Copies back what the unchanged epilogue of optimize! needs (src/solver.jl:167-201): w, w_prev (n+m each),
 * s (m), mu (m) with mu = rho .* (w_prev[n+1:] - s) recovered first.  Any pointer may be NULL. */
COSMO_HIP_API int32_t cosmo_hip_get_iterates(cosmo_hip_handle* h, cosmo_hip_real* w, cosmo_hip_real* w_prev, cosmo_hip_real* s, cosmo_hip_real* mu);
/* sol = [x_tl; nu] of the last KKT solve (ws.sol, src/solver.jl:227-228), length n+m. */
/* Single-launch CG (csrc/cg_persist.hip): out = {enabled for this handle (operator fits one XCD's L2), participating workgroups,
 * persistent launches so far, fallbacks to the multi-kernel path, tickets / barrier arrivals / abort flag of the last launch, LDS
 * doubles per quarter}.  COSMO_HIP_CG_PERSIST=0 / 1 in the environment disables / forces it. */
COSMO_HIP_API int32_t cosmo_hip_cg_persist_stats(cosmo_hip_handle* h, int64_t out[8]);
COSMO_HIP_API int32_t cosmo_hip_get_kkt_solution(cosmo_hip_handle* h, cosmo_hip_real* sol);
/* Assembled reduced operator of the CG solve (csrc/cg_fold.hip): M = P + diag(sigma + d) + Am' rho Am as ONE sparse matrix where the
 * operator split leaves a sparse Am' rho Am (decomposed SDPs), two launches per Krylov iteration instead of three.
 * out = {enabled, nnz of the fully assembled M, rho-weighted terms behind the stored entries, CSR-stream tiles, rows of Am kept FACTORED (partial assembly,
 * round 6: a row of len >= 4 nonzeros costs len^2 assembled entries but 2 len factored ones; 0 = fully assembled), stored entries of [Ms | Ad' rho]}.
 * COSMO_HIP_OP_FOLD=0 in the environment disables the assembly, COSMO_HIP_FOLD_FACTOR=0 the factored rows. */
COSMO_HIP_API int32_t cosmo_hip_fold_stats(cosmo_hip_handle* h, int64_t out[6]);
/* Which Krylov recurrence / kernels the KKT solves of this handle run (valid after cosmo_hip_set_params; a string owned by the library, e.g.
 * "cg: literal recurrence on the assembled operator, two launches per iteration, k_cg_dirM<3, false> + k_cg_upd<false>"): bench.py's
 * config.kkt_solver and roofline.kernel. */
COSMO_HIP_API const char* cosmo_hip_kkt_recurrence(cosmo_hip_handle* h);
/* Statistics of the device loop since set_iterates: out = {admm_iters, kkt_solves, kkt_iters_total,
 * kkt_budget_stalls, spmv_A_calls, spmv_AT_calls, spmv_P_calls, rho_updates}. */
COSMO_HIP_API int32_t cosmo_hip_get_stats(cosmo_hip_handle* h, int64_t out[8]);
/* The automatic rho interval (settings.adaptive_rho_interval == 0, src/solver.jl:244-256) compares the loop's elapsed time with
 * adaptive_rho_fraction * ws.times.setup_time; setup! ends AFTER cosmo_hip_set_params, so its duration is handed over separately (any time
 * before cosmo_hip_optimize; overrides cosmo_hip_params.setup_time). */
COSMO_HIP_API int32_t cosmo_hip_set_setup_time(cosmo_hip_handle* h, double seconds);
/* out = {adaptive_rho_interval in force (0: the automatic rule has not fired yet), iteration at which the automatic rule fixed it or -1}: what
 * the reference writes back into settings.adaptive_rho_interval (src/solver.jl:249-254) */
COSMO_HIP_API int32_t cosmo_hip_get_rho_interval(cosmo_hip_handle* h, int64_t out[2]);

/* ---- measurement hooks (bench.py / rocprof cross-check) ----------------------------------------------- */
/* Times `reps` back-to-back launches of the SpMV kernel `which` (COSMO_HIP_MAT_A/AT/P, or 3 = the fused
 * [P A'] operator kernel of the CG apply) with HIP events on the handle's stream; returns the average
 * seconds per launch and the ALGORITHMIC bytes of one launch. */
COSMO_HIP_API int32_t cosmo_hip_time_spmv(cosmo_hip_handle* h, int32_t which, int32_t reps, double* avg_seconds,
                            double* algorithmic_bytes);
/* Per-kernel-class durations measured with HIP events on the handle's stream.
 * on = 0: off.  on = 1: events around every loop kernel + exact launches.  on = 2: exact launches only (for rocprofv3).
 * "Exact launches": the host synchronises after every Krylov iteration, so no budgeted launch is a guarded no-op and
 * every launch of a kernel does its full work -- per-kernel averages (events or rocprof) are then comparable with the
 * algorithmic bytes of one launch.  Kernel durations are GPU-side and not affected by the host pacing. */
COSMO_HIP_API int32_t cosmo_hip_set_profiling(cosmo_hip_handle* h, int32_t on);
#define COSMO_HIP_NUM_KERNEL_CLASSES 16
COSMO_HIP_API int32_t cosmo_hip_get_kernel_times(cosmo_hip_handle* h, double seconds[COSMO_HIP_NUM_KERNEL_CLASSES],
                                   int64_t launches[COSMO_HIP_NUM_KERNEL_CLASSES]);
COSMO_HIP_API const char* cosmo_hip_kernel_class_name(int32_t k);
/* Jacobi eigensolver diagnostics of the PSD projections: out = {max sweeps used by a single-workgroup solve, sweeps of
 * the last multi-workgroup solve, non-convergence flag, number of PSD cones}. */
COSMO_HIP_API int32_t cosmo_hip_psd_stats(cosmo_hip_handle* h, int64_t out[4]);
/* How PsdCone / PsdConeTriangle are projected (call after set_cones; rebuilds the cone plans; before optimize).  SIGN (default): side <= 16 by a
 * wave-level Jacobi eigensolver, above it the verified matrix-sign iteration on the fp64 matrix cores.  EIGEN: the eigendecomposition-based projection
 * of the reference (src/convexset.jl:163-189, 243-263: syevr! + rank_k_update!) at every side -- Jacobi eigensolvers, X+ = sum_{lambda > 0} lambda v v',
 * nnz_lambda (cosmo_hip_project's psd_rank_out) counted from the eigenvalues themselves.  Complex Hermitian cones keep the sign path. */
enum { COSMO_HIP_PSD_PROJECTION_SIGN = 0, COSMO_HIP_PSD_PROJECTION_EIGEN = 1 };
COSMO_HIP_API int32_t cosmo_hip_set_psd_projection(cosmo_hip_handle* h, int32_t mode);

/* Matrix-sign (polar) PSD path diagnostics: out = {large cones (d > 256), batched cones (64 < d <= 256), tile side of the first
 * large cone, its k-split (1 | 2 | 3 = stream-K), product launches <64,1>, <96,1>, <96,2> or stream-K, batched product launches, matrix products of the main
 * schedule of the last large-cone projection, fallback rounds executed so far, verified projections, products of the last batched
 * projection, steps of the main schedule, unverified projections, projections, max verified error bound in units of 1e-18}. */
COSMO_HIP_API int32_t cosmo_hip_polar_stats(cosmo_hip_handle* h, int64_t out[16]);
/* Opt-in stream-K product kernel of the large cones (COSMO_HIP_POLAR_STREAMK=1; d > 256; k-split reported as 3 by cosmo_hip_polar_stats,
 * its launches under <96,2>):
 * out = {enabled, workgroups per launch of the first large cone, its ticket classes, spin time-outs so far (must stay 0)}. */
COSMO_HIP_API int32_t cosmo_hip_polar_streamk_stats(cosmo_hip_handle* h, int64_t out[4]);
/* The coefficient table of the sign iteration for k_lift lifting steps: abc holds 3 * (*nsteps) doubles (a, b, c per step;
 * NULL = only report *nsteps).  Host function (no device needed): lets a CPU test replay the schedule on scalars. */
COSMO_HIP_API int32_t cosmo_hip_polar_schedule(int32_t k_lift, double* abc, int32_t* nsteps);
/* Measurement hook: `reps` back-to-back launches of the symmetric-product kernel of the sign iteration exactly as the projection
 * launches it (which = 0: first large cone, 1: the whole batch of mid-size cones, Y = U^2 only; 2: the batch's IN-LOOP MIX -- per
 * repetition one Y = U^2 and two alpha A B + beta Cin products with the operands of a step of the iteration, 3 reps launches in all),
 * timed with HIP events on the handle's stream.  Returns the average seconds per launch and the flops one launch performs
 * (2 ts^2 k per upper tile). */
COSMO_HIP_API int32_t cosmo_hip_time_psd_product(cosmo_hip_handle* h, int32_t which, int32_t reps, double* avg_seconds, double* flops);
/* Per-cone lifting depth of the matrix-sign projections (src/convexset.jl:219-263 is what they compute; the depth only steers how many products a
 * projection spends before its a-posteriori verification): out = {adaptive control on, min, max, mean x 1000 of the per-cone depths, d^3-weighted
 * products per projection x 1000 of the batch's last main schedule, failed verifications so far, downward probes so far, projections seen}.
 * OPT-IN: COSMO_HIP_POLAR_ADAPT=1 (measured slower on BASELINE config 5 although it saves 22 % of the products: csrc/psd_polar.hip, PolarPlan::adapt);
 * by default every cone runs the plan's k_lift lifting steps (10 in Float64). */
/* The main schedule of the batched sign iteration as ONE persistent, dependency-driven launch (csrc/psd_polar.hip: k_polar_dataflow; default where every
 * XCD's tile list is at least as long as its workgroup slots, COSMO_HIP_POLAR_DATAFLOW=0|1 forces it): out = {enabled, launches, products per launch,
 * event-timed launches, average seconds per timed launch (HIP events on the handle's stream), matrix flops performed per launch, workgroups, tiles per product}. */
COSMO_HIP_API int32_t cosmo_hip_polar_dataflow_stats(cosmo_hip_handle* h, double out[8]);
COSMO_HIP_API int32_t cosmo_hip_polar_dataflow_reset_timing(cosmo_hip_handle* h);
COSMO_HIP_API int32_t cosmo_hip_polar_depth_stats(cosmo_hip_handle* h, int64_t out[8]);
/* Measurement hook: `reps` Krylov iterations of the reduced CG solve (src/linear_solver/kktsolver_indirect.jl:57-70; IterativeSolvers cg!)
 * exactly as the loop enqueues them -- solve start on the current right-hand side with tolerance 0, then the iterations (captured chain
 * included), HIP events around the iterations only.  The warm start is restored afterwards; the ADMM state is untouched.  Returns the
 * average seconds per Krylov iteration INCLUDING the kernel boundaries between its launches, the algorithmic bytes of one iteration
 * (12 B per nonzero of the operator + row pointers + 8 B per vector element read or written) and the launches per iteration.
 * kkt_kind CG / CG_JACOBI only; call after at least one loop iteration. */
COSMO_HIP_API int32_t cosmo_hip_time_krylov(cosmo_hip_handle* h, int32_t reps, double* avg_seconds, double* algorithmic_bytes, int32_t* launches_per_iteration);

/* ---- clique-sharded projections over the GPUs of one node (one process per GPU, RCCL over xGMI) -----------------------
 * The reference projects the cones of a decomposed SDP serially (src/convexset.jl:885-891).  Here every rank holds the whole
 * problem, runs the identical affine steps, projects only the SOC / PSD cones of its contiguous cone range, and the slices
 * of s are exchanged once per iteration (ncclBroadcast group = all-gather with unequal counts).  No other collective. */
COSMO_HIP_API int32_t cosmo_hip_comm_unique_id(uint8_t id[128]);                       /* ncclGetUniqueId on one rank; host layer distributes it */
COSMO_HIP_API int32_t cosmo_hip_comm_init(cosmo_hip_handle* h, int32_t rank, int32_t nranks, const uint8_t id[128]);
COSMO_HIP_API int32_t cosmo_hip_comm_destroy(cosmo_hip_handle* h);
/* first_cone[nranks+1]: contiguous partition of the cone indices; call after cosmo_hip_set_cones */
COSMO_HIP_API int32_t cosmo_hip_set_cone_shard(cosmo_hip_handle* h, const int64_t* first_cone);
COSMO_HIP_API int32_t cosmo_hip_comm_selftest(cosmo_hip_handle* h);
/* Known-answer test of the all-reduce of `count` reals the row-sharded loop relies on (north_star: "RCCL ... for the residual-norm
 * all-reduce only"; the sums it replaces are src/linear_solver/kktsolver_indirect.jl:52-54 and src/residuals.jl:12-18), through the
 * loop's own code path.  Collective: every rank calls it with the same count.  out = {elements of an exactly representable sum that
 * came back wrong, elements of a fractional sum outside nranks * eps * sum |terms|, FNV-1a hash of the fractional result's bytes (must
 * agree on all ranks: compare it across them), transport (1 RCCL / 2 host-staged), nranks, RCCL version code (0 when host-staged)}.
 * COSMO_HIP_COMM_CORRUPT_RANK=r (test hook) makes rank r contribute 1.001 x its vector to every all-reduce. */
COSMO_HIP_API int32_t cosmo_hip_comm_allreduce_check(cosmo_hip_handle* h, int64_t count, int64_t out[6]);
/* Host-staged communicator for functional tests on a single-GPU host: ranks are processes that may share one device (RCCL
 * refuses that), slices travel through the POSIX shared-memory segment `name` ("/..."; rank 0 creates it).  Same ownership,
 * slices and exchange point as the RCCL path; synchronous, never a performance path.  Call after cosmo_hip_set_problem. */
COSMO_HIP_API int32_t cosmo_hip_comm_init_hostshm(cosmo_hip_handle* h, int32_t rank, int32_t nranks, const char* name);
/* out = {nranks, rank, exchange steps executed with nranks > 1, transport (0 none, 1 RCCL, 2 host-staged)} */
COSMO_HIP_API int32_t cosmo_hip_comm_stats(cosmo_hip_handle* h, int64_t out[4]);
/* ownership without a communicator: project only the SOC / PSD cones cone_lo <= k < cone_hi (testing / custom exchange) */
COSMO_HIP_API int32_t cosmo_hip_set_cone_ownership(cosmo_hip_handle* h, int64_t cone_lo, int64_t cone_hi);

/* ---- row-sharded runs (csrc/rowshard.hip; SURVEY 8e option 2 on a replicated n-side CG) ------------------------------------
 * Replaces the serial cone loop of src/convexset.jl:885-891 AND the row-local parts of admm_x! / admm_w! / the primal residual
 * (src/solver.jl:50-65, src/residuals.jl:2-9): rank r owns the cones first_cone[r] <= k < first_cone[r+1] (all kinds), their rows of A,
 * their columns of A' and the matching slices of b / rho / s / mu / w_s.  The n-vectors and the reduced operator of the CG
 * (src/linear_solver/kktsolver_indirect.jl:57-64) are replicated.  Exchange: ONE all-reduce(sum) of the n-vector A'(rho .* ls_s) per
 * iteration (kktsolver_indirect.jl:52-54) and one of A' mu (+ 2 nranks norms) per residual check -- s is never exchanged.
 * Call on a fully set-up handle: set_problem, set_cones, [scale_ruiz], set_params, comm_init / comm_init_hostshm, then this, then
 * set_iterates (which takes the GLOBAL x0, s0, mu0; get_iterates returns the GLOBAL vectors on every rank).  CG kkt kinds only. */
COSMO_HIP_API int32_t cosmo_hip_set_row_shard(cosmo_hip_handle* h, const int64_t* first_cone);
/* out = {first row, end row, global rows m_g, nnz of this rank's rows of A, its cones, its first cone} */
COSMO_HIP_API int32_t cosmo_hip_row_shard_info(cosmo_hip_handle* h, int64_t out[6]);
/* out = {nranks, rank, collectives of the loop executed with nranks > 1, transport (0 none, 1 RCCL, 2 host-staged), mode (0 none, 1 cone-
 * sharded projections, 2 row-sharded), payload bytes of those collectives, all-reduces of n-vectors, elements of the last all-reduce} */
COSMO_HIP_API int32_t cosmo_hip_comm_stats_ex(cosmo_hip_handle* h, int64_t out[8]);

/* ---- batches of independent problems (BASELINE config 3) ------------------------------------------------------------
 * The reference solves a batch with one optimize!(model) per problem (src/solver.jl:78-203).  Here all problems of a batch
 * (identical n, m and cone structure; data, Box bounds, scalings differ) are solved concurrently, one persistent workgroup
 * per problem running the complete loop src/solver.jl:137-176 with per-problem rho / CG / status.  CG KKT solver only;
 * cones: ZeroSet, Nonnegatives, Box, SecondOrderCone.  Ranks of a multi-GPU job each own a contiguous shard of the batch
 * (no collective). */
typedef struct cosmo_hip_batch cosmo_hip_batch;
COSMO_HIP_API int32_t cosmo_hip_batch_create(cosmo_hip_batch** b, int32_t device_id, int64_t nprob, int64_t n, int64_t m);
COSMO_HIP_API int32_t cosmo_hip_batch_destroy(cosmo_hip_batch* b);
COSMO_HIP_API const char* cosmo_hip_batch_last_error(const cosmo_hip_batch* b);
/* problem k of the batch; arguments as cosmo_hip_set_problem */
COSMO_HIP_API int32_t cosmo_hip_batch_set_problem(cosmo_hip_batch* b, int64_t k, const int64_t* P_colptr, const int64_t* P_rowval,
                                    const cosmo_hip_real* P_nzval, const int64_t* A_colptr, const int64_t* A_rowval,
                                    const cosmo_hip_real* A_nzval, const cosmo_hip_real* q, const cosmo_hip_real* bvec);
/* cone structure shared by all problems; box_l / box_u hold nprob * (#Box rows) entries, problem-major.  Cone kinds of batch mode: ZeroSet,
 * Nonnegatives, Box, SecondOrderCone, PsdCone / PsdConeTriangle of side <= 64 (src/convexset.jl:25-28, 71-74, 100-114, 303-321, 402-412, 844-847)
 * and, through cosmo_hip_batch_set_cones_ex, the exponential / power cones; anything else returns COSMO_HIP_ERR_UNSUPPORTED (one handle per
 * problem serves it) */
COSMO_HIP_API int32_t cosmo_hip_batch_set_cones(cosmo_hip_batch* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                  const cosmo_hip_real* box_l, const cosmo_hip_real* box_u);
/* the same with the per-cone parameter of cosmo_hip_set_cones_ex (alpha of PowerCone / DualPowerCone; NULL = none): additionally
 * ExponentialCone, DualExponentialCone, PowerCone, DualPowerCone (src/convexset.jl:497-779) -- one thread of the problem's workgroup per cone */
COSMO_HIP_API int32_t cosmo_hip_batch_set_cones_ex(cosmo_hip_batch* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                     const cosmo_hip_real* box_l, const cosmo_hip_real* box_u, const cosmo_hip_real* cone_param);
COSMO_HIP_API int32_t cosmo_hip_batch_set_scaling(cosmo_hip_batch* b, int64_t k, const cosmo_hip_real* Dinv, const cosmo_hip_real* Einv, double cinv);
/* The reference's accelerator for every problem of the batch (replaces _make_accelerator!, src/setup.jl:10-16, once per model): the whole
 * accelerated loop of src/solver.jl:140-165 runs inside the problem's persistent workgroup -- acceleration_pre! (update! / accelerate! of the
 * Type-II Anderson accelerator with QR memory, restarted when full), safeguarding with its extra ADMM step (acceleration_post!,
 * src/accelerator_interface.jl:85-116), rho updates and infeasibility checks deferred to the next non-accelerated iteration
 * (update_suggested, src/solver.jl:284-292), IterActivation / AccuracyActivation -- with all decisions per problem on the device.  Call BEFORE
 * cosmo_hip_batch_set_params (the kernel variant and its LDS layout are chosen there); mem <= 16; kind EMPTY / NULL removes it.
 * cosmo_hip_result.iter then counts the safeguarding iterations too (src/solver.jl:196). */
COSMO_HIP_API int32_t cosmo_hip_batch_set_accelerator(cosmo_hip_batch* b, const cosmo_hip_accel_params* p);
/* per problem {accelerated steps, safeguard accepted, safeguard declined, memory restarts, active, safeguarding_iter}: out[6 * nprob] */
COSMO_HIP_API int32_t cosmo_hip_batch_get_accel_stats(cosmo_hip_batch* b, int64_t* out);
/* finalises the batch (uploads, classify_constraints!, set_rho_vec! per problem) */
COSMO_HIP_API int32_t cosmo_hip_batch_set_params(cosmo_hip_batch* b, const cosmo_hip_params* p);
COSMO_HIP_API int32_t cosmo_hip_batch_get_rho_classes(cosmo_hip_batch* b, int64_t k, int32_t* cls /* m */);
/* x0: nprob*n, s0 / mu0: nprob*m, problem-major; NULL = zeros (src/solver.jl:128-129 per problem) */
COSMO_HIP_API int32_t cosmo_hip_batch_set_iterates(cosmo_hip_batch* b, const cosmo_hip_real* x0, const cosmo_hip_real* s0, const cosmo_hip_real* mu0);
/* optimize! for every problem; results has nprob entries */
COSMO_HIP_API int32_t cosmo_hip_batch_optimize(cosmo_hip_batch* b, cosmo_hip_result* results);
/* n_iters more loop bodies on every undecided problem, with the residual checks / adaptive-rho checks of the schedule but WITHOUT the
 * infeasibility certificates (src/solver.jl:326-349): only cosmo_hip_batch_optimize cuts the persistent launch at the iterations the
 * reference tests at.  A measurement / stepping entry: an infeasible problem driven through it runs on undecided.  with_init != 0 runs
 * the init step first */
COSMO_HIP_API int32_t cosmo_hip_batch_iterate(cosmo_hip_batch* b, int64_t n_iters, int32_t with_init);
/* per problem {ADMM iterations, KKT solves, Krylov iterations in total}: out[3 * nprob] (measurement; the counters the reference keeps in
 * IndirectReducedKKTSolver.iteration_counter / multiplications, src/linear_solver/kktsolver_indirect.jl:32,56) */
COSMO_HIP_API int32_t cosmo_hip_batch_get_counters(cosmo_hip_batch* b, int64_t* out);
/* which kernel the batch runs (after set_params; measurement / tests): out = {form: 0 streaming, 1 LDS image, 2 register kernel <512, 1, 2>,
 * 3 register kernel <512, 2, 4>; sliced image 0 / 1; dynamic LDS bytes per workgroup; P held in registers 0 / 1; registers per thread and
 * scratch bytes per thread of that kernel instantiation as the loaded code object reports them (> 0 scratch: it spills); its static LDS bytes;
 * bit 0: length-sorted compute assignment in the Krylov loop, bit 1: rows of >= 64 entries handed to a whole wave (cooperative long-row passes)}
 * (ABI 1004: out grew from 4 to 8 entries) */
COSMO_HIP_API int32_t cosmo_hip_batch_kernel_info(cosmo_hip_batch* b, int64_t out[8]);
COSMO_HIP_API int32_t cosmo_hip_batch_get_iterates(cosmo_hip_batch* b, int64_t k, cosmo_hip_real* w, cosmo_hip_real* w_prev, cosmo_hip_real* s, cosmo_hip_real* mu);

/* ---- batches of problems of DIFFERENT structure (csrc/batch_group.hip) ------------------------------------------------------------------
 * The reference's batch is a loop over arbitrary models (src/solver.jl:78).  A group takes every problem with ITS OWN (n, m, cones), partitions
 * them at set_params time into classes of identical structure (dimensions, cone types / dimensions / parameters; data, Box bounds and scalings
 * differ freely), builds one cosmo_hip_batch per class -- so every class keeps the persistent kernel specialised for its structure -- and
 * cosmo_hip_batch_group_optimize runs all classes concurrently (one HIP stream + host thread per class).  Call order as for a batch:
 * create, set_problem / set_cones [/ set_scaling] for every k, [set_accelerator,] set_params, [set_iterates,] optimize, get_iterates.
 * A class the batch kernels refuse (PSD side > 64, MINRES, adaptive_rho_interval = 0) is solved through one single-problem handle per member instead;
 * what no path of the library takes (an unknown cone type; user-defined cones need their callbacks: one handle per problem) is the group's error,
 * naming the problem. */
typedef struct cosmo_hip_batch_group cosmo_hip_batch_group;
COSMO_HIP_API int32_t cosmo_hip_batch_group_create(cosmo_hip_batch_group** g, int32_t device_id, int64_t nprob);
COSMO_HIP_API int32_t cosmo_hip_batch_group_destroy(cosmo_hip_batch_group* g);
COSMO_HIP_API const char* cosmo_hip_batch_group_last_error(const cosmo_hip_batch_group* g);
/* problem k, n x n P and m x n A as cosmo_hip_set_problem */
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_problem(cosmo_hip_batch_group* g, int64_t k, int64_t n, int64_t m, const int64_t* P_colptr, const int64_t* P_rowval,
                                          const cosmo_hip_real* P_nzval, const int64_t* A_colptr, const int64_t* A_rowval,
                                          const cosmo_hip_real* A_nzval, const cosmo_hip_real* q, const cosmo_hip_real* bvec);
/* cones of problem k as cosmo_hip_set_cones_ex; box_l / box_u = the Box rows of THIS problem */
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_cones(cosmo_hip_batch_group* g, int64_t k, int64_t ncones, const int32_t* type, const int64_t* dim,
                                        const cosmo_hip_real* box_l, const cosmo_hip_real* box_u, const cosmo_hip_real* cone_param);
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_scaling(cosmo_hip_batch_group* g, int64_t k, const cosmo_hip_real* Dinv, const cosmo_hip_real* Einv, double cinv);
/* ... with D, E, c themselves (as cosmo_hip_set_scaling_full): members that run on their own handles test their certificates with exactly these */
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_scaling_full(cosmo_hip_batch_group* g, int64_t k, const cosmo_hip_real* D, const cosmo_hip_real* Dinv, const cosmo_hip_real* E,
                                               const cosmo_hip_real* Einv, double c, double cinv);
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_accelerator(cosmo_hip_batch_group* g, const cosmo_hip_accel_params* p);
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_params(cosmo_hip_batch_group* g, const cosmo_hip_params* p);
/* number of structure classes; class_of[k] and mode_of[k] for every problem (nprob entries each, may be NULL): mode 0 = the class runs on a
 * persistent batch kernel, 1 = its structure is outside the batch kernels (PSD side > 64, a MINRES solver kind, ...) and every member is solved
 * through its own single-problem handle, concurrently with the batch classes */
COSMO_HIP_API int32_t cosmo_hip_batch_group_class_info(cosmo_hip_batch_group* g, int64_t* nclasses, int64_t* class_of, int64_t* mode_of);
/* cosmo_hip_get_rho_interval for problem k of a group (members with the automatic interval run on their own handles): out as there */
COSMO_HIP_API int32_t cosmo_hip_batch_group_get_rho_interval(cosmo_hip_batch_group* g, int64_t k, int64_t out[2]);
/* out = {worker threads of the last cosmo_hip_batch_group_optimize (a bounded pool: COSMO_HIP_GROUP_WORKERS, default 32), its jobs (one per MERGED SET of
 * one-problem classes -- all of them run in one host loop whose every launch covers them, workgroup c reading the descriptor of class c --, one per other batch
 * class, one per member that runs on its own handle), classes, problems, classes that ran inside a merged set} */
COSMO_HIP_API int32_t cosmo_hip_batch_group_run_info(cosmo_hip_batch_group* g, int64_t out[5]);
/* warm start of problem k (n, m, m entries; NULL = zeros); problems never set start from zero (src/solver.jl:128-129) */
COSMO_HIP_API int32_t cosmo_hip_batch_group_set_iterates(cosmo_hip_batch_group* g, int64_t k, const cosmo_hip_real* x0, const cosmo_hip_real* s0, const cosmo_hip_real* mu0);
/* optimize! for every problem; results has nprob entries in the caller's order */
COSMO_HIP_API int32_t cosmo_hip_batch_group_optimize(cosmo_hip_batch_group* g, cosmo_hip_result* results);
COSMO_HIP_API int32_t cosmo_hip_batch_group_get_iterates(cosmo_hip_batch_group* g, int64_t k, cosmo_hip_real* w, cosmo_hip_real* w_prev, cosmo_hip_real* s, cosmo_hip_real* mu);
COSMO_HIP_API int32_t cosmo_hip_batch_group_get_counters(cosmo_hip_batch_group* g, int64_t* out /* 3 * nprob */);
COSMO_HIP_API int32_t cosmo_hip_batch_group_get_accel_stats(cosmo_hip_batch_group* g, int64_t* out /* 6 * nprob */);

#ifdef __cplusplus
}
#endif
#endif /* COSMO_HIP_H */
