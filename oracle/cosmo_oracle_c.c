/* cosmo_oracle_c.c -- TEST / BASELINE INFRASTRUCTURE, not part of the product (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it).  A plain-C restatement of the reference's ADMM loop (COSMO.jl v0.8.11) for the configurations whose
 * cones are ZeroSet / Nonnegatives / Box / SecondOrderCone / PsdCone / PsdConeTriangle and whose KKT solver is the CG reduced solver --
 * i.e. all five BASELINE configs (since round 4 optionally with the reference's default AndersonAccelerator and its safeguarding) -- compiled with gcc -O3 so that the CPU number quoted next to the GPU number comes from compiled
 * code, as the (Julia) reference's would.  The PSD projections call LAPACK ?syevr and BLAS ?syrk exactly as the reference does
 * (src/convexset.jl:163-189, 243-263) through function pointers handed in by the loader (SciPy's bundled OpenBLAS: the image has no
 * liblapack to link against).
 * It follows oracle/cosmo_oracle.py line by line (which is pinned on the reference's goldens) and is itself pinned against it in
 * tests/test_oracle_c.py (same iteration counts, iterates to 1e-9).  Setup (scaling, classification, rho vector) stays in the
 * NumPy oracle; this file is the loop only:
 *   src/solver.jl:137-176 (loop), :7-21 admm_z!, :32-56 admm_x!, :62-65 admm_w!, :24-26 recover_mu!, :242-282 rho rules,
 *   :303-323 check_termination! (residual part); src/convexset.jl:25-28,71-74,844-847 (Zero / Nonneg / Box), :100-114 (SecondOrderCone),
 *   :219-263, 303-321, 402-412, 432-442, 462-472 (PSD cones), :885-891 (serial cone loop); src/residuals.jl:1-153;
 *   src/parameters.jl:3-92; src/linear_solver/kktsolver_indirect.jl:36-88 with IterativeSolvers v0.9 cg! (restated, see the
 *   header of the NumPy oracle).  SpMVs are Julia's CSC kernels: A x scatters column by column, A'y is a dot per column.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  double sigma, alpha, rho, eps_abs, eps_rel, tol_constant, tol_exponent;
  double rho_min, rho_max, rho_eq_over_rho_ineq, adaptive_rho_tolerance, cinv;
  int64_t max_iter, adaptive_rho_max_adaptions;
  int32_t check_termination, adaptive_rho, adaptive_rho_interval, unscale;
  /* accelerator (src/settings.jl:136-138): accel != 0 = AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory, NoRegularizer} */
  int32_t accel, acc_mem, acc_min_mem, safeguard;
  int64_t acc_start_iter;            /* 2 = ImmediateActivation, k = IterActivation(k) */
  double safeguard_tol;
} oc_params;

typedef struct {
  int32_t status;            /* 1 Solved, 2 Max_iter_reached, 3 Unsolved */
  int32_t n_rho_updates;
  int64_t iter, cg_iters_total;
  double cost, r_prim, r_dual, max_norm_prim, max_norm_dual, rho, iter_time;
  int64_t safeguarding_iter, num_accelerated;   /* iter = loop index + safeguarding_iter (total_iter, src/solver.jl:196) */
} oc_result;

/* Element type of the loop: double (COSMO.Model{Float64}) by default, float with -DOC_FLOAT (the Float32 instantiation that checks
 * libcosmo_hip_f32.so; every operation of the loop then rounds to Float32 as Julia's Float32 broadcasts do -- the file compiles
 * without -Wdouble-promotion warnings).  Settings and result scalars stay double in both builds. */
#ifdef OC_FLOAT
typedef float oc_real;
#define RSQRT(x) sqrtf(x)
#define RFABS(x) fabsf(x)
#define RFMAX(a, b) fmaxf((a), (b))
#define RFMIN(a, b) fminf((a), (b))
#define RPOW(a, b) powf((a), (b))
#else
typedef double oc_real;
#define RSQRT(x) sqrt(x)
#define RFABS(x) fabs(x)
#define RFMAX(a, b) fmax((a), (b))
#define RFMIN(a, b) fmin((a), (b))
#define RPOW(a, b) pow((a), (b))
#endif
#define R(x) ((oc_real)(x))
#define RINF ((oc_real)INFINITY)


typedef struct { int64_t nr, nc; const int64_t* p; const int64_t* i; const oc_real* x; } csc;

static void mul(const csc* M, const oc_real* v, oc_real* y) {          /* y = M v : column scatter (Julia mul!(y, A, x)) */
  for (int64_t r = 0; r < M->nr; ++r) y[r] = R(0.0);
  for (int64_t j = 0; j < M->nc; ++j) { const oc_real vj = v[j]; for (int64_t k = M->p[j]; k < M->p[j + 1]; ++k) y[M->i[k]] += M->x[k] * vj; }
}
static void mulT(const csc* M, const oc_real* v, oc_real* y) {         /* y = M' v : one dot per column (mul!(y, A', x)) */
  for (int64_t j = 0; j < M->nc; ++j) { oc_real s = R(0.0); for (int64_t k = M->p[j]; k < M->p[j + 1]; ++k) s += M->x[k] * v[M->i[k]]; y[j] = s; }
}
static oc_real nrm2(const oc_real* v, int64_t n) { oc_real s = R(0.0); for (int64_t i = 0; i < n; ++i) s += v[i] * v[i]; return RSQRT(s); }
static oc_real dot(const oc_real* a, const oc_real* b, int64_t n) { oc_real s = R(0.0); for (int64_t i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static oc_real amax(oc_real acc, oc_real v) { const oc_real a = RFABS(v); return (a > acc || a != a) ? a : acc; }
static oc_real maxn(oc_real a, oc_real b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }   /* max that keeps a NaN (Julia) */

typedef struct {
  int64_t n, m;
  csc P, A;
  const oc_real *q, *b, *Dinv, *Einv;
  const int32_t* cls;          /* rho class per row: 0 inequality, 1 equality, 2 loose */
  const int32_t* kind;         /* per row: 0 free (unused), 1 zero, 2 nonneg, 3 box */
  const oc_real *bl, *bu;       /* per row (only read on box rows) */
  oc_real *rho, *tn, *tm, *tm2; /* work */
} prob;

/* ---- cones that are projected slice by slice after the row-wise ones (src/convexset.jl:885-891: a serial loop over the cones) ---- */
enum { CK_SOC = 1, CK_PSD_TRIANGLE = 2, CK_PSD_SQUARE = 3 };
typedef void (*syevr_fn)(char* jobz, char* range, char* uplo, int* n, oc_real* a, int* lda, oc_real* vl, oc_real* vu, int* il, int* iu, oc_real* abstol,
                         int* m, oc_real* w, oc_real* z, int* ldz, int* isuppz, oc_real* work, int* lwork, int* iwork, int* liwork, int* info);
typedef void (*syrk_fn)(char* uplo, char* trans, int* n, int* k, oc_real* alpha, oc_real* a, int* lda, oc_real* beta, oc_real* c, int* ldc);
typedef struct {
  int64_t ncones; const int32_t* kind; const int64_t* off; const int64_t* dim;
  syevr_fn syevr; syrk_fn syrk;
  oc_real *X, *Z, *wv, *work; int *iwork, *isuppz; int lwork, liwork;      /* PsdBlasWorkspace (convexset.jl:129-161) */
  int64_t* rank_out; int32_t* branch_out;                                   /* per cone, may be NULL */
} cone_ctx;

static int64_t isqrt64(int64_t v) { int64_t r = (int64_t)sqrt((double)v); while (r * r > v) --r; while ((r + 1) * (r + 1) <= v) ++r; return r; }
static int64_t psd_side(int32_t kind, int64_t dim) { return kind == CK_PSD_SQUARE ? isqrt64(dim) : (isqrt64(1 + 8 * dim) - 1) / 2; }   /* :372 */

/* _project! (convexset.jl:219-241): w, Z = syevr('V','A','U', X) ; rank_k_update! (:243-263): X = sum_{lambda > 0} lambda z z' (upper triangle) */
static int psd_project_dense(cone_ctx* cx, int d, int64_t* nnz_out) {
  char jobz = 'V', range = 'A', uplo = 'U', trans = 'N';
  int n = d, m_found = 0, info = 0, il = 0, iu = 0;
  oc_real vl = R(0.0), vu = R(0.0), abstol = R(-1.0);
  cx->syevr(&jobz, &range, &uplo, &n, cx->X, &n, &vl, &vu, &il, &iu, &abstol, &m_found, cx->wv, cx->Z, &n, cx->isuppz, cx->work, &cx->lwork, cx->iwork,
            &cx->liwork, &info);
  if (info != 0) return info;
  int nnz = 0;
  for (int j = 0; j < d; ++j)
    if (cx->wv[j] > R(0.0)) { nnz += 1; const oc_real sq = RSQRT(cx->wv[j]); oc_real* z = cx->Z + (size_t)j * d; for (int i = 0; i < d; ++i) z[i] = z[i] * sq; }   /* :250-254 */
  if (nnz > 0) {
    oc_real one = R(1.0), zero = R(0.0);
    cx->syrk(&uplo, &trans, &n, &nnz, &one, cx->Z + (size_t)(d - nnz) * d, &n, &zero, cx->X, &n);       /* :258-261: the positive pairs are the LAST columns */
  } else {
    memset(cx->X, 0, sizeof(oc_real) * (size_t)d * d);
  }
  *nnz_out = nnz;
  return 0;
}

static int project_cones(cone_ctx* cx, oc_real* s) {
  const oc_real isq2 = R(1.0) / RSQRT(R(2.0)), sq2 = RSQRT(R(2.0));
  for (int64_t c = 0; c < cx->ncones; ++c) {
    oc_real* x = s + cx->off[c];
    const int64_t dim = cx->dim[c];
    if (cx->kind[c] == CK_SOC) {                                          /* convexset.jl:100-114 */
      int br = 0;
      if (dim > 0) {
        const oc_real t = x[0];
        const oc_real nx = nrm2(x + 1, dim - 1);
        if (nx <= t) br = 0;
        else if (nx <= -t) { br = 1; for (int64_t i = 0; i < dim; ++i) x[i] = R(0.0); }
        else { br = 2; x[0] = (nx + t) / R(2.0); const oc_real f = (nx + t) / (R(2.0) * nx); for (int64_t i = 1; i < dim; ++i) x[i] = f * x[i]; }
      }
      if (cx->branch_out) cx->branch_out[c] = br;
      continue;
    }
    int64_t nnz = 0;
    if (dim == 1) {                                                       /* :307-308, 404-405 */
      x[0] = (x[0] != x[0]) ? x[0] : ((x[0] > R(0.0)) ? x[0] : R(0.0));
      nnz = x[0] > R(0.0);
    } else if (cx->kind[c] == CK_PSD_TRIANGLE) {                          /* :402-412 */
      const int d = (int)psd_side(CK_PSD_TRIANGLE, dim);
      int64_t k = 0;
      for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i, ++k) cx->X[(size_t)j * d + i] = (i == j) ? x[k] : isq2 * x[k];      /* populate_upper_triangle! (:432-442) */
      const int rc = psd_project_dense(cx, d, &nnz);
      if (rc) return rc;
      k = 0;
      for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i, ++k) x[k] = (i == j) ? cx->X[(size_t)j * d + i] : sq2 * cx->X[(size_t)j * d + i];   /* extract_upper_triangle! (:462-472) */
    } else {                                                              /* PsdCone, :303-321 */
      const int d = (int)psd_side(CK_PSD_SQUARE, dim);
      for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) cx->X[(size_t)j * d + i] = (x[(size_t)j * d + i] + x[(size_t)i * d + j]) / R(2.0);   /* symmetrize_upper! (algebra.jl:201-208) */
      const int rc = psd_project_dense(cx, d, &nnz);
      if (rc) return rc;
      for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) { const oc_real v = cx->X[(size_t)j * d + i]; x[(size_t)j * d + i] = v; x[(size_t)i * d + j] = v; }   /* :316-318 */
    }
    if (cx->rank_out) cx->rank_out[c] = nnz;
  }
  return 0;
}

/* reduced_mul! (kktsolver_indirect.jl:57-64): y = P x + sigma x + A'(rho .* (A x)) */
static void reduced_mul(const prob* W, oc_real sigma, const oc_real* x, oc_real* y, oc_real* tmp_m, oc_real* tmp_n) {
  mul(&W->A, x, tmp_m);
  for (int64_t i = 0; i < W->m; ++i) tmp_m[i] *= W->rho[i];
  mulT(&W->A, tmp_m, tmp_n);
  for (int64_t j = 0; j < W->n; ++j) tmp_n[j] = sigma * x[j] + tmp_n[j];
  mul(&W->P, x, y);
  for (int64_t j = 0; j < W->n; ++j) y[j] = tmp_n[j] + y[j];
}

static void make_rho(const prob* W, const oc_params* p, oc_real rho) {   /* set_rho_vec! / update_rho_vec! (parameters.jl:3-13,75-92) */
  for (int64_t i = 0; i < W->m; ++i)
    W->rho[i] = (W->cls[i] == 1) ? R(p->rho_eq_over_rho_ineq) * rho : (W->cls[i] == 2 ? R(p->rho_min) : rho);
}


/* ---- direct KKT path: the reference's DEFAULT solver QdldlKKTSolver (src/linear_solver/kktsolver.jl:285-320) -----------------------------
 * K = [P + sigma I, A'; A, -diag(1 ./ rho)] (upper triangle, CSC; assemble_kkt_triangle :175-250), `qdldl(K)` = a fill-reducing symmetric permutation
 * (AMD.jl) + QDLDL's LDL' (elimination tree, then the up-looking numeric factorisation), `solve!` = copy + permute + L \ . , D \ . , L' \ . + permute back
 * (:310-313), `update_rho!` = rewrite the last m diagonal entries of K and refactor numerically (:316-320).  QDLDL.jl (compat "0.4.1",
 * /root/reference/Project.toml:32) and AMD.jl ("0.4, 0.5") are NOT vendored in the reference tree: what follows restates the published QDLDL
 * algorithm (Stellato et al., OSQP, Math. Prog. Comp. 2020, section 5.1 / the qdldl C sources it ships: QDLDL_etree, QDLDL_factor, QDLDL_Lsolve,
 * QDLDL_Ltsolve) -- PARITY UNPINNED like the Krylov packages; anchored on a dense solve of the same system (tests/test_oracle_direct_kkt.py).  The
 * permutation is an input (the Python side takes a minimum-degree ordering; any symmetric permutation gives the same solution).
 * Used ONLY as bench.py's cpu_baseline.direct_kkt leg (the CPU path north_star names) and by its test. */
typedef struct {
  int64_t N;                       /* n + m */
  const int64_t *Kp, *Ki;          /* upper triangle of the PERMUTED K, CSC, row indices ascending, diagonal LAST in every column */
  oc_real* Kx;
  const int64_t* perm;             /* permuted position k holds original index perm[k] */
  const int64_t* rho_pos;          /* position in Kx of the diagonal entry of original row n + i (i < m) */
  int64_t *Lp, *Li, *etree, *Lnz, *iwork; unsigned char* bwork;
  oc_real *Lx, *D, *Dinv, *fwork, *xw;
  int64_t nnzL, positive_D, n_factor;
  double factor_time, solve_time; int64_t n_solve;
} ldl_t;
#define LDL_UNKNOWN (-1)
/* QDLDL_etree: elimination tree + column counts of L; returns nnz(L), -1 (entry below the diagonal / empty column) or -2 (more than `cap` nonzeros) */
static int64_t ldl_etree(int64_t n, const int64_t* Ap, const int64_t* Ai, int64_t* work, int64_t* Lnz, int64_t* etree, int64_t cap) {
  for (int64_t i = 0; i < n; ++i) { work[i] = 0; Lnz[i] = 0; etree[i] = LDL_UNKNOWN; if (Ap[i] == Ap[i + 1]) return -1; }
  int64_t sum = 0;
  for (int64_t j = 0; j < n; ++j) {
    work[j] = j;
    for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) {
      int64_t i = Ai[p];
      if (i > j) return -1;
      while (work[i] != j) {
        if (etree[i] == LDL_UNKNOWN) etree[i] = j;
        Lnz[i] += 1; sum += 1;
        work[i] = j;
        i = etree[i];
      }
    }
    if (cap > 0 && sum > cap) return -2;
  }
  return sum;
}
/* QDLDL_factor: up-looking LDL'; row k of L is the solution of a sparse triangular system whose pattern is read off the elimination tree */
static int ldl_factor(ldl_t* F) {
  const int64_t n = F->N; const int64_t *Ap = F->Kp, *Ai = F->Ki; const oc_real* Ax = F->Kx;
  int64_t *Lp = F->Lp, *Li = F->Li; oc_real *Lx = F->Lx, *D = F->D, *Dinv = F->Dinv;
  unsigned char* yMarkers = F->bwork; int64_t* yIdx = F->iwork; int64_t* elimBuffer = F->iwork + n; int64_t* LNext = F->iwork + 2 * n; oc_real* yVals = F->fwork;
  int64_t positive = 0;
  Lp[0] = 0;
  for (int64_t i = 0; i < n; ++i) { Lp[i + 1] = Lp[i] + F->Lnz[i]; yMarkers[i] = 0; yVals[i] = R(0.0); D[i] = R(0.0); LNext[i] = Lp[i]; }
  D[0] = Ax[0];
  if (D[0] == R(0.0)) return -1;
  if (D[0] > R(0.0)) positive += 1;
  Dinv[0] = R(1.0) / D[0];
  for (int64_t k = 1; k < n; ++k) {
    int64_t nnzY = 0;
    for (int64_t i = Ap[k]; i < Ap[k + 1]; ++i) {
      const int64_t bidx = Ai[i];
      if (bidx == k) { D[k] = Ax[i]; continue; }
      yVals[bidx] = Ax[i];
      int64_t next = bidx;
      if (!yMarkers[next]) {
        yMarkers[next] = 1;
        elimBuffer[0] = next;
        int64_t nnzE = 1;
        next = F->etree[bidx];
        while (next != LDL_UNKNOWN && next < k) {
          if (yMarkers[next]) break;
          yMarkers[next] = 1;
          elimBuffer[nnzE++] = next;
          next = F->etree[next];
        }
        while (nnzE) yIdx[nnzY++] = elimBuffer[--nnzE];
      }
    }
    for (int64_t i = nnzY - 1; i >= 0; --i) {
      const int64_t cidx = yIdx[i];
      const int64_t tmpIdx = LNext[cidx];
      const oc_real yc = yVals[cidx];
      for (int64_t j = Lp[cidx]; j < tmpIdx; ++j) yVals[Li[j]] -= Lx[j] * yc;
      Li[tmpIdx] = k;
      Lx[tmpIdx] = yc * Dinv[cidx];
      D[k] -= yc * Lx[tmpIdx];
      LNext[cidx] += 1;
      yVals[cidx] = R(0.0);
      yMarkers[cidx] = 0;
    }
    if (D[k] == R(0.0)) return -1;
    if (D[k] > R(0.0)) positive += 1;
    Dinv[k] = R(1.0) / D[k];
  }
  F->positive_D = positive;
  return 0;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int ldl_refactor(ldl_t* F) { const double t0 = now_s(); const int rc = ldl_factor(F); F->factor_time += now_s() - t0; F->n_factor += 1; return rc; }
/* QDLDL.solve!: x <- P' (L' \ (D \ (L \ (P x)))) */
static void ldl_solve(ldl_t* F, const oc_real* rhs, oc_real* out) {
  const double t0 = now_s();
  const int64_t n = F->N; oc_real* x = F->xw;
  for (int64_t k = 0; k < n; ++k) x[k] = rhs[F->perm[k]];
  for (int64_t i = 0; i < n; ++i) { const oc_real v = x[i]; for (int64_t j = F->Lp[i]; j < F->Lp[i + 1]; ++j) x[F->Li[j]] -= F->Lx[j] * v; }
  for (int64_t i = 0; i < n; ++i) x[i] *= F->Dinv[i];
  for (int64_t i = n - 1; i >= 0; --i) { oc_real v = x[i]; for (int64_t j = F->Lp[i]; j < F->Lp[i + 1]; ++j) v -= F->Lx[j] * x[F->Li[j]]; x[i] = v; }
  for (int64_t k = 0; k < n; ++k) out[F->perm[k]] = x[k];
  F->solve_time += now_s() - t0; F->n_solve += 1;
}
static void ldl_free(ldl_t* F) { free(F->Lp); free(F->Li); free(F->etree); free(F->Lnz); free(F->iwork); free(F->bwork); free(F->Lx); free(F->D); free(F->Dinv); free(F->fwork); free(F->xw); }
/* symbolic analysis + workspace; 0 ok, 1 out of memory, 4 K is not upper triangular with a full diagonal, 5 nnz(L) > cap */
static int ldl_setup(ldl_t* F, int64_t cap) {
  const int64_t n = F->N;
  F->Lp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1)); F->etree = (int64_t*)malloc(sizeof(int64_t) * (size_t)n); F->Lnz = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  F->iwork = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)n); F->bwork = (unsigned char*)malloc((size_t)n);
  F->D = (oc_real*)malloc(sizeof(oc_real) * (size_t)n); F->Dinv = (oc_real*)malloc(sizeof(oc_real) * (size_t)n); F->fwork = (oc_real*)malloc(sizeof(oc_real) * (size_t)n);
  F->xw = (oc_real*)malloc(sizeof(oc_real) * (size_t)n);
  if (!F->Lp || !F->etree || !F->Lnz || !F->iwork || !F->bwork || !F->D || !F->Dinv || !F->fwork || !F->xw) return 1;
  const int64_t nnzL = ldl_etree(n, F->Kp, F->Ki, F->iwork, F->Lnz, F->etree, cap);
  if (nnzL == -1) return 4;
  if (nnzL == -2) return 5;
  F->nnzL = nnzL;
  F->Li = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nnzL + 1)); F->Lx = (oc_real*)malloc(sizeof(oc_real) * (size_t)(nnzL + 1));
  if (!F->Li || !F->Lx) return 1;
  return 0;
}
/* nnz(L) of the permuted upper triangle alone (etree pass), -2 if it exceeds `cap` (> 0): the feasibility probe of bench.py for BASELINE config 2 */
int64_t cosmo_oracle_c_ldl_nnz(int64_t N, const int64_t* Kp, const int64_t* Ki, int64_t cap) {
  int64_t* w = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)N);
  if (!w) return -3;
  const int64_t r = ldl_etree(N, Kp, Ki, w, w + N, w + 2 * N, cap);
  free(w);
  return r;
}

/* ncones slice cones (SOC / PSD; kind per row 0 on their rows) with LAPACK / BLAS entry points `syevr`, `syrk` (NULL when there is no PSD cone);
 * rank_out / branch_out: nnz_lambda / SOC branch id of the LAST projection per cone (NULL to skip); proj_time_out: seconds spent in the cone loop */
/* ---- AndersonAccelerator (COSMOAccelerators.jl, external to the reference tree: restated from the published algorithm exactly as
 * oracle/cosmo_oracle.py: class AndersonAccelerator does -- update!: f = x - g, G_j = g - g_last, v = f - f_last, modified Gram-Schmidt
 * column by column; accelerate!: eta = R \ (Q' f), rejected if R is singular / not finite or ||eta||_2 > 1e4, else g -= G eta) ---- */
typedef struct {
  int64_t dim; int mem, min_mem, iter, init_phase, success;
  int64_t num_accelerated, num_restarts;
  oc_real *G, *Q, *Rm, *f, *f_last, *g_last, *eta, *v, *rhs;
} aa_t;
static void aa_restart(aa_t* a) {                                     /* CA.restart! -> empty_history! */
  memset(a->G, 0, sizeof(oc_real) * (size_t)a->dim * a->mem); memset(a->Q, 0, sizeof(oc_real) * (size_t)a->dim * a->mem);
  memset(a->Rm, 0, sizeof(oc_real) * (size_t)a->mem * a->mem);
  memset(a->f, 0, sizeof(oc_real) * (size_t)a->dim); memset(a->f_last, 0, sizeof(oc_real) * (size_t)a->dim); memset(a->g_last, 0, sizeof(oc_real) * (size_t)a->dim);
  a->iter = 0; a->init_phase = 1;
}
static void aa_update(aa_t* a, const oc_real* g, const oc_real* x) {
  const int64_t N = a->dim;
  for (int64_t i = 0; i < N; ++i) a->f[i] = x[i] - g[i];
  if (a->init_phase) {
    memcpy(a->g_last, g, sizeof(oc_real) * (size_t)N); memcpy(a->f_last, a->f, sizeof(oc_real) * (size_t)N);
    a->init_phase = 0;
    return;
  }
  int j = a->iter % a->mem;
  if (j == 0 && a->iter != 0) {                                       /* RestartedMemory */
    memset(a->G, 0, sizeof(oc_real) * (size_t)N * a->mem); memset(a->Q, 0, sizeof(oc_real) * (size_t)N * a->mem);
    memset(a->Rm, 0, sizeof(oc_real) * (size_t)a->mem * a->mem);
    a->iter = 0; a->num_restarts += 1;
  }
  oc_real* Gj = a->G + (size_t)j * N; oc_real* v = a->v;
  for (int64_t i = 0; i < N; ++i) { Gj[i] = g[i] - a->g_last[i]; v[i] = a->f[i] - a->f_last[i]; }
  memcpy(a->g_last, g, sizeof(oc_real) * (size_t)N); memcpy(a->f_last, a->f, sizeof(oc_real) * (size_t)N);
  for (int i = 0; i < j; ++i) {                                       /* qr!: modified Gram-Schmidt */
    const oc_real* Qi = a->Q + (size_t)i * N;
    const oc_real r = dot(Qi, v, N);
    a->Rm[(size_t)j * a->mem + i] = r;                                /* R[i, j], column-major */
    for (int64_t e = 0; e < N; ++e) v[e] = v[e] - r * Qi[e];
  }
  const oc_real nv = nrm2(v, N);
  a->Rm[(size_t)j * a->mem + j] = nv;
  oc_real* Qj = a->Q + (size_t)j * N;
  for (int64_t e = 0; e < N; ++e) Qj[e] = v[e] / nv;
  a->iter += 1;
}
static void aa_accelerate(aa_t* a, oc_real* g) {
  const int64_t N = a->dim;
  a->success = 0;
  const int l = a->iter < a->mem ? a->iter : a->mem;
  if (l < a->min_mem) return;
  for (int c = 0; c < l; ++c) a->rhs[c] = dot(a->Q + (size_t)c * N, a->f, N);
  for (int c = 0; c < l; ++c)
    for (int r = 0; r <= c; ++r) { const oc_real x = a->Rm[(size_t)c * a->mem + r]; if (!(RFABS(x) < (oc_real)INFINITY)) return; }       /* Inf or NaN */
  for (int c = 0; c < l; ++c) if (a->Rm[(size_t)c * a->mem + c] == R(0.0)) return;      /* trtrs info > 0: exactly singular */
  oc_real nrm = R(0.0);
  for (int i = l - 1; i >= 0; --i) {                                  /* back substitution */
    oc_real sacc = R(0.0);
    for (int k = i + 1; k < l; ++k) sacc += a->Rm[(size_t)k * a->mem + i] * a->eta[k];
    a->eta[i] = (a->rhs[i] - sacc) / a->Rm[(size_t)i * a->mem + i];
    nrm += a->eta[i] * a->eta[i];
  }
  nrm = RSQRT(nrm);
  if (!(nrm <= R(1e4))) return;                                       /* also NaN */
  for (int64_t i = 0; i < N; ++i) { oc_real sacc = R(0.0); for (int c = 0; c < l; ++c) sacc += a->G[(size_t)c * N + i] * a->eta[c]; g[i] -= sacc; }   /* g -= G[:, 1:l] eta */
  a->num_accelerated += 1;
  a->success = 1;
}

static int32_t run_loop(ldl_t* dk, int64_t n, int64_t m, const int64_t* Pp, const int64_t* Pi, const oc_real* Px, const int64_t* Ap, const int64_t* Ai,
                                 const oc_real* Ax, const oc_real* q, const oc_real* b, const oc_real* Dinv, const oc_real* Einv, const int32_t* cls,
                                 const int32_t* kind, const oc_real* bl, const oc_real* bu, const oc_params* prm, const oc_real* rho_vec0,
                                 oc_real* x_io, oc_real* s_io, oc_real* mu_io, oc_real* rho_updates_out, int32_t rho_updates_cap, oc_result* res,
                                 int64_t ncones, const int32_t* ckind, const int64_t* coff, const int64_t* cdim, void* syevr, void* syrk,
                                 int64_t* rank_out, int32_t* branch_out, double* proj_time_out) {
  cone_ctx cx;
  memset(&cx, 0, sizeof cx);
  cx.ncones = ncones; cx.kind = ckind; cx.off = coff; cx.dim = cdim; cx.syevr = (syevr_fn)syevr; cx.syrk = (syrk_fn)syrk;
  cx.rank_out = rank_out; cx.branch_out = branch_out;
  int dmax = 0;
  for (int64_t c = 0; c < ncones; ++c) if (ckind[c] != CK_SOC && cdim[c] > 1) { const int d = (int)psd_side(ckind[c], cdim[c]); if (d > dmax) dmax = d; }
  void* psd_buf = NULL;
  if (dmax > 0) {
    if (!cx.syevr || !cx.syrk) return 2;
    /* workspace query (LAPACK.syevr! with lwork = -1, convexset.jl:140-156) */
    char jobz = 'V', range = 'A', uplo = 'U';
    int nq = dmax, mq = 0, info = 0, il = 0, iu = 0, lq = -1, liq = -1, iwq = 0, isq[2] = {0, 0};
    oc_real vl = R(0.0), vu = R(0.0), abstol = R(-1.0), wq = R(0.0), dummy = R(0.0);
    cx.syevr(&jobz, &range, &uplo, &nq, &dummy, &nq, &vl, &vu, &il, &iu, &abstol, &mq, &dummy, &dummy, &nq, isq, &wq, &lq, &iwq, &liq, &info);
    cx.lwork = (int)wq; cx.liwork = iwq;
    if (info != 0 || cx.lwork < 26 * dmax) cx.lwork = 26 * dmax;
    if (cx.liwork < 10 * dmax) cx.liwork = 10 * dmax;
    const size_t nd = (size_t)dmax * dmax;
    psd_buf = malloc(sizeof(oc_real) * (2 * nd + (size_t)dmax + (size_t)cx.lwork) + sizeof(int) * ((size_t)cx.liwork + 2 * (size_t)dmax + 2));
    if (!psd_buf) return 1;
    cx.X = (oc_real*)psd_buf; cx.Z = cx.X + nd; cx.wv = cx.Z + nd; cx.work = cx.wv + dmax;
    cx.iwork = (int*)(cx.work + cx.lwork); cx.isuppz = cx.iwork + cx.liwork;
  }
  double proj_time = 0.0;
  prob W;
  W.n = n; W.m = m;
  W.P.nr = n; W.P.nc = n; W.P.p = Pp; W.P.i = Pi; W.P.x = Px;
  W.A.nr = m; W.A.nc = n; W.A.p = Ap; W.A.i = Ai; W.A.x = Ax;
  W.q = q; W.b = b; W.Dinv = Dinv; W.Einv = Einv; W.cls = cls; W.kind = kind; W.bl = bl; W.bu = bu;
  const oc_params p = *prm;
  /* the settings are Float64 in the ABI; the loop computes with them in its element type (COSMO.Settings{T}) */
  const oc_real p_sigma = R(p.sigma), p_alpha = R(p.alpha), p_rho = R(p.rho), p_eps_abs = R(p.eps_abs), p_eps_rel = R(p.eps_rel), p_tol_constant = R(p.tol_constant),
                p_tol_exponent = R(p.tol_exponent), p_rho_min = R(p.rho_min), p_rho_max = R(p.rho_max), p_adaptive_rho_tolerance = R(p.adaptive_rho_tolerance), p_cinv = R(p.cinv);
  const int64_t N = n + m;
  oc_real* buf = (oc_real*)calloc((size_t)(4 * N + 6 * m + 9 * n + 16), sizeof(oc_real));
  if (!buf) return 1;
  oc_real* w = buf; oc_real* w_prev = w + N; oc_real* ls = w_prev + N; oc_real* sol = ls + N;
  oc_real* s = sol + N; oc_real* mu = s + m; oc_real* s_tl = mu + m; oc_real* tmp_m = s_tl + m; oc_real* y2 = tmp_m + m; W.rho = y2 + m;
  oc_real* prev = W.rho + m; oc_real* tmp_n = prev + n; oc_real* y1 = tmp_n + n; oc_real* r = y1 + n; oc_real* u = r + n; oc_real* c = u + n;
  oc_real* rd = c + n; oc_real* rt = rd + n; oc_real* tn2 = rt + n;
  memcpy(W.rho, rho_vec0, sizeof(oc_real) * (size_t)m);
  oc_real rho = p_rho;
  int n_rho = 1;
  if (rho_updates_out && rho_updates_cap > 0) rho_updates_out[0] = rho;
  int64_t iteration_counter = 1, cg_total = 0;
  for (int64_t j = 0; j < n; ++j) w[j] = x_io[j];                        /* solver.jl:128-129 */
  for (int64_t i = 0; i < m; ++i) { w[n + i] = (R(1.0) / W.rho[i]) * mu_io[i] + s_io[i]; s[i] = s_io[i]; mu[i] = mu_io[i]; }
  int status = 0;
  oc_real cost = RINF, r_prim = RINF, r_dual = RINF, mnp = R(0.0), mnd = R(0.0);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);

#define SOLVE_AND_W()                                                                                                         \
  {                                                                                                                           \
    for (int64_t j = 0; j < n; ++j) ls[j] = p_sigma * w[j] - q[j];                              /* solver.jl:50 */            \
    for (int64_t i = 0; i < m; ++i) ls[n + i] = (b[i] - R(2.0) * s[i]) + w[n + i];                 /* :51 */                     \
    if (dk) {                                                        /* solve!(::QdldlKKTSolver, sol, ls) (kktsolver.jl:310-313) */         \
      ldl_solve(dk, ls, sol);                                                                                                 \
    } else {                                                                                                                  \
    for (int64_t i = 0; i < m; ++i) y2[i] = W.rho[i] * ls[n + i];                               /* kktsolver_indirect.jl:52 */\
    mulT(&W.A, y2, y1);                                                                                                       \
    for (int64_t j = 0; j < n; ++j) y1[j] = y1[j] + ls[j];                                                                    \
    const oc_real nb = nrm2(y1, n);                                                                                            \
    const oc_real tolk = p_tol_constant / RPOW((oc_real)iteration_counter, p_tol_exponent);                                      \
    const oc_real abstol = nb > R(0.0) ? tolk / nb : RINF;                                                                    \
    /* cg! v0.9: warm start from `prev`, maxiter = n */                                                                       \
    reduced_mul(&W, p_sigma, prev, c, tmp_m, tmp_n);                                                                          \
    for (int64_t j = 0; j < n; ++j) { r[j] = y1[j] - c[j]; u[j] = R(0.0); }                                                      \
    oc_real residual = nrm2(r, n), prev_res = R(1.0);                                                                             \
    int64_t it_cg = 0;                                                                                                        \
    while (it_cg < n && !(residual <= abstol)) {                                                                              \
      const oc_real beta = (residual * residual) / (prev_res * prev_res);                                                      \
      for (int64_t j = 0; j < n; ++j) u[j] = r[j] + beta * u[j];                                                              \
      reduced_mul(&W, p_sigma, u, c, tmp_m, tmp_n);                                                                           \
      const oc_real al = (residual * residual) / dot(u, c, n);                                                                 \
      for (int64_t j = 0; j < n; ++j) { prev[j] += al * u[j]; r[j] -= al * c[j]; }                                            \
      prev_res = residual; residual = nrm2(r, n); ++it_cg;                                                                    \
    }                                                                                                                         \
    cg_total += it_cg;                                                                                                        \
    for (int64_t j = 0; j < n; ++j) sol[j] = prev[j];                                                                         \
    mul(&W.A, prev, tmp_m);                                                                     /* :81-83 */                  \
    for (int64_t i = 0; i < m; ++i) sol[n + i] = (tmp_m[i] - ls[n + i]) * W.rho[i];                                           \
    iteration_counter += 1;                                                                                                   \
    }                                                                                                                         \
    for (int64_t i = 0; i < m; ++i) s_tl[i] = (R(2.0) * s[i] - w[n + i]) - sol[n + i] / W.rho[i];  /* solver.jl:55 */            \
    for (int64_t j = 0; j < n; ++j) w[j] = w[j] + p_alpha * (sol[j] - w[j]);                    /* :63 */                     \
    for (int64_t i = 0; i < m; ++i) w[n + i] = w[n + i] + p_alpha * (s_tl[i] - s[i]);           /* :64 */                     \
  }

  /* residuals of (x = w_prev[1:n], s, mu): calculate_result_info! (residuals.jl:30-93), optionally unscaled */
#define RESIDUALS(UNSCALE)                                                                                                    \
  {                                                                                                                           \
    const oc_real* xx = w_prev;                                                                                                \
    mul(&W.A, xx, tmp_m);                                                                                                     \
    oc_real rp = R(0.0), mp = R(0.0), a1 = R(0.0), a2 = R(0.0), a3 = R(0.0);                                                                  \
    for (int64_t i = 0; i < m; ++i) {                                                                                         \
      const oc_real e = (UNSCALE) ? Einv[i] : R(1.0);                                                                             \
      rp = amax(rp, ((tmp_m[i] + s[i]) - b[i]) * e);                                                                          \
      a1 = amax(a1, tmp_m[i] * e); a2 = amax(a2, s[i] * e); a3 = amax(a3, b[i] * e);                                          \
    }                                                                                                                         \
    mp = maxn(maxn(a1, a2), a3);   /* Julia's max keeps a NaN */                                                                                              \
    mul(&W.P, xx, rd); mulT(&W.A, mu, rt);                                                                                    \
    oc_real rdn = R(0.0), b1 = R(0.0), b2 = R(0.0), b3 = R(0.0);                                                                           \
    const oc_real ci = (UNSCALE) ? p_cinv : R(1.0);                                                                               \
    for (int64_t j = 0; j < n; ++j) {                                                                                         \
      const oc_real dj = (UNSCALE) ? Dinv[j] : R(1.0);                                                                            \
      rdn = amax(rdn, (((rd[j] + q[j]) - rt[j]) * dj) * ci);                                                                  \
      b1 = amax(b1, (rd[j] * dj) * ci); b2 = amax(b2, (q[j] * dj) * ci); b3 = amax(b3, (rt[j] * dj) * ci);                    \
    }                                                                                                                         \
    r_prim = rp; r_dual = rdn; mnp = mp; mnd = maxn(maxn(b1, b2), b3);                                                        \
  }

  SOLVE_AND_W()                                                           /* init step (solver.jl:137-138) */
  int64_t it = 0, sg = 0;
  int rho_update_due = 0;
  aa_t aa;
  memset(&aa, 0, sizeof aa);
  oc_real* aa_buf = NULL;
  int acc_active = 0;
  if (p.accel) {                                                          /* _make_accelerator! (setup.jl:10-16): mem = min(mem, dim) */
    aa.dim = N; aa.mem = (int)((int64_t)p.acc_mem < N ? p.acc_mem : N); if (aa.mem < 1) aa.mem = 1;
    aa.min_mem = p.acc_min_mem; aa.init_phase = 1;
    const size_t Nm = (size_t)N * aa.mem;
    aa_buf = (oc_real*)calloc(2 * Nm + (size_t)aa.mem * aa.mem + 4 * (size_t)N + 2 * (size_t)aa.mem + 8, sizeof(oc_real));
    if (!aa_buf) { free(buf); free(psd_buf); return 1; }
    aa.G = aa_buf; aa.Q = aa.G + Nm; aa.Rm = aa.Q + Nm; aa.f = aa.Rm + (size_t)aa.mem * aa.mem; aa.f_last = aa.f + N; aa.g_last = aa.f_last + N;
    aa.v = aa.g_last + N; aa.eta = aa.v + N; aa.rhs = aa.eta + aa.mem;
  }
  while (it + sg < p.max_iter) {                                          /* :140 */
    it += 1;
    if (p.accel) {                                                        /* acceleration_pre! (accelerator_interface.jl:58-76) */
      if (!acc_active && it >= p.acc_start_iter) acc_active = 1;
      if (acc_active) { aa_update(&aa, w, w_prev); aa_accelerate(&aa, w); }
    }
    memcpy(w_prev, w, sizeof(oc_real) * (size_t)N);                       /* :151 */
    for (int64_t i = 0; i < m; ++i) {                                    /* admm_z! (:7-21) */
      oc_real v = w[n + i];
      switch (kind[i]) {
        case 1: v = R(0.0); break;
        case 2: v = (v != v) ? v : ((v > R(0.0)) ? v : R(0.0)); break;
        case 3: v = (v < bl[i]) ? bl[i] : ((v > bu[i]) ? bu[i] : v); break;
        default: break;
      }
      s[i] = v;
    }
    if (ncones > 0) {                                                    /* project!(s, C): the slice cones (convexset.jl:885-891) */
      struct timespec p0, p1;
      clock_gettime(CLOCK_MONOTONIC, &p0);
      if (project_cones(&cx, s) != 0) { free(buf); free(psd_buf); free(aa_buf); return 3; }
      clock_gettime(CLOCK_MONOTONIC, &p1);
      proj_time += (double)(p1.tv_sec - p0.tv_sec) + 1e-9 * (double)(p1.tv_nsec - p0.tv_nsec);
    }
    if (p.adaptive_rho && p.adaptive_rho_interval > 0 && (it % p.adaptive_rho_interval) == 0 && (n_rho - 1) < p.adaptive_rho_max_adaptions)
      rho_update_due = 1;
    if (rho_update_due && !(p.accel && aa.success)) {                    /* apply_rho_adaptation_rules! (:242-282); update_suggested (:284-292) */
      rho_update_due = 0;
      for (int64_t i = 0; i < m; ++i) mu[i] = W.rho[i] * (w_prev[n + i] - s[i]);
      oc_real sr_p = r_prim, sr_d = r_dual, smp = mnp, smd = mnd;
      RESIDUALS(0)
      const oc_real rp = r_prim / (mnp + R(1e-10)), rdd = r_dual / (mnd + R(1e-10));
      r_prim = sr_p; r_dual = sr_d; mnp = smp; mnd = smd;
      oc_real new_rho = rho * RSQRT(rp / (rdd + R(1e-10)));
      if (new_rho == new_rho) new_rho = RFMIN(RFMAX(new_rho, p_rho_min), p_rho_max);   /* Julia's min(max(NaN, .), .) is NaN: no update */
      if (new_rho > p_adaptive_rho_tolerance * rho || new_rho < (R(1.0) / p_adaptive_rho_tolerance) * rho) {
        rho = new_rho;
        make_rho(&W, &p, rho);
        if (dk) {                                                          /* update_rho! (kktsolver.jl:316-320): new diagonal + numeric refactorisation */
          for (int64_t i = 0; i < m; ++i) dk->Kx[dk->rho_pos[i]] = -R(1.0) / W.rho[i];
          if (ldl_refactor(dk) != 0) { free(buf); free(psd_buf); free(aa_buf); return 6; }
        }
        if (rho_updates_out && n_rho < rho_updates_cap) rho_updates_out[n_rho] = rho;
        n_rho += 1;
        if (p.accel) aa_restart(&aa);                                   /* :272-275: the ADMM operator changed */
        for (int64_t i = 0; i < m; ++i) w[n + i] = (R(1.0) / W.rho[i]) * mu[i] + s[i];
      }
    }
    SOLVE_AND_W()
    if (p.accel && acc_active && aa.success && p.safeguard) {            /* acceleration_post! (accelerator_interface.jl:85-117) */
      const oc_real nrm_tol = nrm2(aa.f, N) * R(p.safeguard_tol);
      for (int64_t i = 0; i < N; ++i) aa.f[i] = w_prev[i] - w[i];          /* compute_accelerated_res_norm! (:123-126) */
      if (nrm2(aa.f, N) > nrm_tol) {
        memcpy(w_prev, aa.g_last, sizeof(oc_real) * (size_t)N); memcpy(w, aa.g_last, sizeof(oc_real) * (size_t)N);   /* reset_accelerated_vector! */
        for (int64_t i = 0; i < m; ++i) {
          oc_real v = w[n + i];
          switch (kind[i]) {
            case 1: v = R(0.0); break;
            case 2: v = (v != v) ? v : ((v > R(0.0)) ? v : R(0.0)); break;
            case 3: v = (v < bl[i]) ? bl[i] : ((v > bu[i]) ? bu[i] : v); break;
            default: break;
          }
          s[i] = v;
        }
        if (ncones > 0 && project_cones(&cx, s) != 0) { free(buf); free(psd_buf); free(aa_buf); return 3; }
        SOLVE_AND_W()
        sg += 1;
      }
    }
    if ((it % p.check_termination) == 0 || it == 1) {                    /* check_termination! (:306-323) */
      for (int64_t i = 0; i < m; ++i) mu[i] = W.rho[i] * (w_prev[n + i] - s[i]);
      RESIDUALS(p.unscale)
      mul(&W.P, w_prev, tn2);
      cost = p_cinv * (R(0.5) * dot(tn2, w_prev, n) + dot(q, w_prev, n));
      if (RFABS(cost) > R(1e20)) { status = 3; break; }
      if (r_prim < p_eps_abs + p_eps_rel * mnp && r_dual < p_eps_abs + p_eps_rel * mnd) { status = 1; break; }
    }
  }
  for (int64_t i = 0; i < m; ++i) mu[i] = W.rho[i] * (w_prev[n + i] - s[i]);    /* :167 */
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (it + sg == p.max_iter) { RESIDUALS(p.unscale) status = 2; }               /* :173-176 */
  for (int64_t j = 0; j < n; ++j) x_io[j] = w_prev[j];
  memcpy(s_io, s, sizeof(oc_real) * (size_t)m);
  memcpy(mu_io, mu, sizeof(oc_real) * (size_t)m);
  res->status = status; res->n_rho_updates = n_rho; res->iter = it + sg; res->cg_iters_total = cg_total;    /* total_iter (:196) */
  res->safeguarding_iter = sg; res->num_accelerated = aa.num_accelerated;
  free(aa_buf);
  res->cost = cost; res->r_prim = r_prim; res->r_dual = r_dual; res->max_norm_prim = mnp; res->max_norm_dual = mnd; res->rho = rho;
  res->iter_time = (oc_real)(t1.tv_sec - t0.tv_sec) + R(1e-9) * (oc_real)(t1.tv_nsec - t0.tv_nsec);
  free(buf);
  free(psd_buf);
  if (proj_time_out) *proj_time_out = proj_time;
  return 0;
}

int32_t cosmo_oracle_c_run_cones(int64_t n, int64_t m, const int64_t* Pp, const int64_t* Pi, const oc_real* Px, const int64_t* Ap, const int64_t* Ai,
                                 const oc_real* Ax, const oc_real* q, const oc_real* b, const oc_real* Dinv, const oc_real* Einv, const int32_t* cls,
                                 const int32_t* kind, const oc_real* bl, const oc_real* bu, const oc_params* prm, const oc_real* rho_vec0,
                                 oc_real* x_io, oc_real* s_io, oc_real* mu_io, oc_real* rho_updates_out, int32_t rho_updates_cap, oc_result* res,
                                 int64_t ncones, const int32_t* ckind, const int64_t* coff, const int64_t* cdim, void* syevr, void* syrk,
                                 int64_t* rank_out, int32_t* branch_out, double* proj_time_out) {
  return run_loop(NULL, n, m, Pp, Pi, Px, Ap, Ai, Ax, q, b, Dinv, Einv, cls, kind, bl, bu, prm, rho_vec0, x_io, s_io, mu_io, rho_updates_out, rho_updates_cap, res,
                  ncones, ckind, coff, cdim, syevr, syrk, rank_out, branch_out, proj_time_out);
}

/* The same loop with the reference's default direct KKT solver.  Kp / Ki / Kx: upper triangle of the symmetrically permuted KKT matrix (Kx is
 * overwritten by rho updates), perm[k] = original index at permuted position k, rho_pos[i] = position in Kx of the diagonal entry of row n + i.
 * ldl_stats_out = {nnz(L), factor seconds (all factorisations), number of factorisations, solve seconds (all solves), number of solves, positive
 * entries of D (inertia: must equal n, kktsolver.jl:304), setup seconds (etree)}.  nnz_cap > 0: give up (return 5) if nnz(L) exceeds it. */
int32_t cosmo_oracle_c_run_cones_direct(int64_t n, int64_t m, const int64_t* Pp, const int64_t* Pi, const oc_real* Px, const int64_t* Ap, const int64_t* Ai,
                                        const oc_real* Ax, const oc_real* q, const oc_real* b, const oc_real* Dinv, const oc_real* Einv, const int32_t* cls,
                                        const int32_t* kind, const oc_real* bl, const oc_real* bu, const oc_params* prm, const oc_real* rho_vec0,
                                        oc_real* x_io, oc_real* s_io, oc_real* mu_io, oc_real* rho_updates_out, int32_t rho_updates_cap, oc_result* res,
                                        int64_t ncones, const int32_t* ckind, const int64_t* coff, const int64_t* cdim, void* syevr, void* syrk,
                                        int64_t* rank_out, int32_t* branch_out, double* proj_time_out,
                                        const int64_t* Kp, const int64_t* Ki, oc_real* Kx, const int64_t* perm, const int64_t* rho_pos, int64_t nnz_cap,
                                        double* ldl_stats_out) {
  ldl_t F;
  memset(&F, 0, sizeof F);
  F.N = n + m; F.Kp = Kp; F.Ki = Ki; F.Kx = Kx; F.perm = perm; F.rho_pos = rho_pos;
  const double t0 = now_s();
  int rc = ldl_setup(&F, nnz_cap);
  const double t_sym = now_s() - t0;
  if (rc == 0 && ldl_refactor(&F) != 0) rc = 6;
  if (rc == 0 && F.positive_D != n) rc = 7;                               /* "Objective function is not convex." (kktsolver.jl:304) */
  if (rc == 0)
    rc = run_loop(&F, n, m, Pp, Pi, Px, Ap, Ai, Ax, q, b, Dinv, Einv, cls, kind, bl, bu, prm, rho_vec0, x_io, s_io, mu_io, rho_updates_out, rho_updates_cap, res,
                  ncones, ckind, coff, cdim, syevr, syrk, rank_out, branch_out, proj_time_out);
  if (ldl_stats_out) {
    ldl_stats_out[0] = (double)F.nnzL; ldl_stats_out[1] = F.factor_time; ldl_stats_out[2] = (double)F.n_factor; ldl_stats_out[3] = F.solve_time;
    ldl_stats_out[4] = (double)F.n_solve; ldl_stats_out[5] = (double)F.positive_D; ldl_stats_out[6] = t_sym;
  }
  ldl_free(&F);
  return rc;
}

/* the row-cone-only entry point of rounds 1-2 (BASELINE configs 1 and 2) */
int32_t cosmo_oracle_c_run(int64_t n, int64_t m, const int64_t* Pp, const int64_t* Pi, const oc_real* Px, const int64_t* Ap, const int64_t* Ai,
                           const oc_real* Ax, const oc_real* q, const oc_real* b, const oc_real* Dinv, const oc_real* Einv, const int32_t* cls,
                           const int32_t* kind, const oc_real* bl, const oc_real* bu, const oc_params* prm, const oc_real* rho_vec0,
                           oc_real* x_io, oc_real* s_io, oc_real* mu_io, oc_real* rho_updates_out, int32_t rho_updates_cap, oc_result* res) {
  return cosmo_oracle_c_run_cones(n, m, Pp, Pi, Px, Ap, Ai, Ax, q, b, Dinv, Einv, cls, kind, bl, bu, prm, rho_vec0, x_io, s_io, mu_io, rho_updates_out,
                                  rho_updates_cap, res, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
}
