// internal.h -- shared declarations of libcosmo_hip (gfx950 only; no portability layers).
//
// Device data layout (all resident in HBM for the life of the handle):
//   sparse matrices : CSR, int32 indices, fp64 values; four copies: A (m x n), A' (n x m), P (n x n) and the
//                     row-merged operator [P | A'] (n x (n+m)) that the CG apply and the dual residual stream once
//   vectors         : w, w_prev (n+m) ; s, mu, s_tl, rho, tmp_m, ls_s, nu (m) ; ls_x, x_tl(=CG iterate), r, u, c,
//                     rhs (n) ; Dinv (n), Einv (m)
//   control block   : one `Ctl` struct in device memory; every data-dependent decision of the loop (CG
//                     convergence, rho adaptation, termination) is taken on the device and recorded there, so the
//                     host can enqueue whole iterations without synchronising.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/cosmo_hip.h"

// ---- the scalar type of this build ---------------------------------------------------------------------------------
// libcosmo_hip.so: real = double (COSMO.Model{Float64}).  libcosmo_hip_f32.so: the same sources with -DCOSMO_HIP_REAL_FLOAT,
// real = float (COSMO.Model{Float32}, src/types.jl:348).  Every data array, every device scalar and all kernel arithmetic is
// `real`; what stays `double` on purpose: wall-clock times, the settings struct of the ABI (converted at the launch sites) and
// host-side bookkeeping that is not part of the iteration's arithmetic.
typedef cosmo_hip_real real;
#ifdef COSMO_HIP_REAL_FLOAT
typedef float2 real2;
#define make_real2 make_float2
#define REAL_EPS 1.1920928955078125e-07f
#define REAL_MAX 3.402823466e+38f
#define REAL_IS_FLOAT 1
#else
typedef double2 real2;
#define make_real2 make_double2
#define REAL_EPS 2.220446049250313e-16
#define REAL_MAX 1.7976931348623157e308
#define REAL_IS_FLOAT 0
#endif
// literal of the build's scalar type: R(2.0) * x must not promote a float expression to double (Julia's Float32 broadcasts round
// every operation to Float32)
#define R(x) ((real)(x))

#define COSMO_BS 256            // threads per workgroup for streaming kernels (4 waves of 64)
#define COSMO_NNZ_PER_BLOCK 2048 // CSR-stream: nonzeros staged in LDS per row block (16 KB of products; measured best of 512..4096)
#define COSMO_MAX_PARTIALS 2048 // upper bound on workgroups that emit reduction partials
#define COSMO_NSLOTS 8          // partial-reduction slots

// ---- device control block -------------------------------------------------------------------------------------
struct Ctl {
  int halt;          // != 0: every loop kernel returns immediately (status decided, stall, or error)
  int status;        // COSMO_HIP_* solver status decided on the device
  int stalled;       // the Krylov budget of the current solve ran out before convergence
  int error;         // COSMO_HIP_ERR_* raised on the device
  int cg_done;       // current solve has converged
  int cg_k;          // Krylov iterations performed in the current solve
  int cg_k_max;      // max over solves since the host last reset it (drives the launch budget)
  int rho_changed;   // set by the adaptation kernel, consumed by the rho-apply kernel
  int n_rho_updates; // length(ws.rho_updates)
  int cg_kd;         // iteration index of the k_cg_dirM in flight, published for the k_cg_upd behind it (device-side index of the captured chain)
  long long iter;    // ADMM iterations completed
  long long solves;  // KKT solves completed (iteration_counter - 1)
  long long kkt_iters_total;
  real resv[2];    // CG residual norms, indexed by iteration parity
  real tol;        // absolute tolerance of the current solve
  real rhs_norm;
  real rho;        // scalar rho (ws.rho)
  real r_prim, r_dual, max_norm_prim, max_norm_dual, cost;
  real minres[16]; // MINRES scalar recurrences, two parity slots of 8 (see minres.hip)
  real udotc_slot; // <v_curr, v_next> of the current MINRES iteration
  real sr_gamma[2], sr_alpha[2];   // single-reduction CG (cg_sr.hip): r'r and alpha of the last two iterations, by parity
  int sr_k[2];       // single-reduction CG, captured chain: the launch of parity p reads its iteration index from sr_k[p] and publishes k + 1 in
                     // sr_k[p ^ 1] (cg_k is rewritten by workgroup 0 WHILE the other workgroups of the same launch would read it)
  real rho_updates[COSMO_HIP_MAX_RHO_UPDATES];
};

// ---- CSR matrix on the device -----------------------------------------------------------------------------------
struct CsrDev {
  int nrows = 0, ncols = 0;
  long long nnz = 0;
  int* rowptr = nullptr;  // nrows+1
  int* col = nullptr;     // nnz
  real* val = nullptr;  // nnz
  int* split = nullptr;   // nrows (merged operator only): index where the A' part of the row starts
  int* rb = nullptr;      // nb+1 row-block boundaries of the CSR-stream schedule
  int nb = 0;             // number of row blocks
  int grid = 0;           // workgroups launched (<= COSMO_MAX_PARTIALS), each loops over row blocks
  int split_col = 0;      // merged operator: columns >= split_col gather from the second vector
  int xcd_affine = 0;     // XCD-affine tile order in the one-tile-per-workgroup kernels (device_utils.h: tile_of_block)
};

// assembled reduced operator of the CG solve (cg_fold.hip; also applied by the single-reduction CG of cg_sr.hip)
struct FoldPlan {
  int slots = 8;            // nonzero slots per thread of k_cg_dirM (tile <= slots * 256)
  CsrDev M;                 // n x n; M.val is rewritten by k_fold_refresh
  real* base = nullptr;   // nnz(M): P_ij (0 where P has no entry)
  int* drow = nullptr;      // nnz(M): row index for diagonal entries, -1 otherwise
  int* tptr = nullptr;      // nnz(M)+1: terms of entry p are [tptr[p], tptr[p+1])
  int* trow = nullptr;      // term -> row of Am
  real* tprod = nullptr;  // term -> a_ki * a_kj
  // tile-major PADDED copy of (col, val) for k_cg_dirM (round 4): tile k's nonzeros start at k * cap, so the workgroup requests them from its
  // block index alone, TOGETHER with the tile descriptor and the partial sums instead of one memory round trip behind the descriptor
  int* pcol = nullptr;      // nb * cap (padding: column 0)
  real* pval = nullptr;   // nb * cap (padding: 0); rewritten by k_fold_refresh next to M.val
  int* ppos = nullptr;      // nnz(M): position of entry p in pval
  int cap = 0;              // slots per tile (= slots * 256); 0: no padded copy (size-heuristic tiles of large operators)
  int* dpos = nullptr;      // n: index of row i's diagonal entry in M.val
  real* dinv = nullptr;   // n: 1 / M_ii, the Jacobi preconditioner of the opt-in PCG (refreshed with the values)
  long long nterms = 0;
  // captured chain of speculative Krylov iterations (cg_fold.hip: fold_enqueue_iterations)
  void* chain = nullptr;    // hipGraphExec_t
  void* chain_cf = nullptr; // the same chain with check_first = 1 (iterations expected to be no-ops)
  int chain_len = 0;        // Krylov iterations per launch of the chain
  int chain_off = 0;        // COSMO_HIP_CG_GRAPH=0
  // PARTIAL ASSEMBLY (round 6): rows of Am with >= 4 nonzeros (`dense` rows: a row of len nonzeros costs len^2 assembled entries but 2 len factored ones) stay
  // FACTORED: M = Ms + Ad' diag(rho_d) Ad.  The stored matrix is the row-merged [Ms | Ad' rho_d] (n x (n + nd), split_col = n); the columns >= n gather from
  // the records tt[kd] = {(Ad r)_kd, (Ad u_prev)_kd} exactly as the columns < n gather from {r, u}.  k_cg_updF forms (Ad r_new) as a FRESH product from the
  // complete vectors of the iteration -- sum a_kj (r_j - alpha c_j), the owners' own expression -- for which the {r, u} records are double-buffered by
  // iteration parity (the owners write the new records while the dense rows gather the old ones).  (A first version advanced (Ad r) by linearity,
  // tr -= alpha (Ad c): a recurrence whose absolute error stays at eps |Ad r_0| while r shrinks -- 5e-7 trajectory deviations at a 1e-10 stopping threshold.)
  int nd = 0;               // dense rows kept factored (0: fully assembled)
  long long nnz_full = 0;   // nonzeros the FULLY assembled operator has (the unit of the bench's algorithmic bytes)
  CsrDev Ad;                // the dense rows (nd x n, values without rho) with their own CSR-stream tiles
  real2* tt = nullptr;    // nd records {Ad r, Ad u_prev}
  real* tcur = nullptr;   // nd: Ad u of the iteration in flight (written by k_cg_dirM, stored into the record by k_cg_updF)
  real* tx = nullptr;     // nd: Ad x (solve start)
  // the same for the one-launch single-reduction recurrence (cg_sr.hip: k_sr_M); chain_len is even there (records / partials alternate by parity)
  void* sr_chain = nullptr;
  void* sr_chain_cf = nullptr;
  int sr_chain_len = 0;
};

struct HostCsr {  // host staging of a CSR matrix (0-based)
  int nrows = 0, ncols = 0;
  std::vector<int> rowptr, col;
  std::vector<real> val;
  std::vector<int> split;
};

enum KernelClass {
  KC_Z = 0, KC_SOC, KC_PSD, KC_RHS, KC_SPMV_AT, KC_SPMV_A, KC_OP_APPLY, KC_CG_DIR, KC_CG_UPD, KC_TAIL,
  KC_CHK_PRIM, KC_CHK_DUAL, KC_CHK_FINAL, KC_RHO_APPLY, KC_MINRES_VEC, KC_OTHER
};

struct ConeTable {  // host copy of the composite set
  std::vector<int32_t> type;
  std::vector<int64_t> dim, off;
  int64_t nbox_rows = 0;
  std::vector<real> box_l, box_u;
  std::vector<real> param;       // per cone: alpha of the power cones
};

struct PsdPlan;  // psd.hip

struct CustomCone {   // custom.hip: a user subtype of AbstractConvexCone, projected on the host by the user's callback
  long long cone = 0, off = 0, dim = 0, host_off = 0;
  cosmo_hip_project_fn project = nullptr;
  cosmo_hip_cone_test_fn in_dual = nullptr, in_pol_recc = nullptr;
  void* user = nullptr;
};

struct cosmo_hip_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool destroyed = false;
  // sizes
  long long n = 0, m = 0;
  bool have_problem = false, have_cones = false, have_params = false, have_iterates = false;
  int psd_mode = 0;                // cosmo_hip_set_psd_projection: 0 = verified matrix-sign iteration above side 16 (default), 1 = eigendecomposition (Jacobi) at every side
  // matrices
  CsrDev A, AT, P, PT;
  // CG operator split (api.hip: build_op_split): rows of A with exactly one nonzero contribute a DIAGONAL to A' rho A;
  // Am = the other rows (compact), PTm = [P | Am'], op_diag[j] = sum_{single rows i in column j} rho_i a_i^2, rho_m = rho on Am's rows
  bool op_split = false;
  CsrDev Am, PTm;
  int* op_mrow = nullptr;          // Am row -> row of A
  int *op_sc_ptr = nullptr, *op_sc_row = nullptr;
  real *op_sc_a2 = nullptr, *op_diag = nullptr, *op_rho_m = nullptr;
  long long op_nsingle = 0;
  // assembled reduced operator M = P + diag(sigma + d) + Am' rho_m Am (cg_fold.hip): two launches per Krylov iteration
  bool op_fold = false;
  void* fold = nullptr;           // FoldPlan
  // data vectors
  real *q = nullptr, *b = nullptr, *rho = nullptr, *Dinv = nullptr, *Einv = nullptr, *Dscale = nullptr, *Escale = nullptr;
  real *inf_dy = nullptr, *inf_dx = nullptr, *inf_adx = nullptr;   // infeasibility work vectors (infeas.hip)
  int* inf_flags = nullptr;
  real cinv = 1.0;
  bool has_scaling = false;
  bool P_symmetric = true;       // set by set_problem; the device Ruiz scaling requires it
  // cones
  ConeTable cones;
  uint32_t* meta = nullptr;       // per row: kind (2 bits) | box index << 2
  real *box_l = nullptr, *box_u = nullptr;
  int* rho_cls = nullptr;         // per row rho class 0/1/2
  std::vector<int32_t> rho_cls_host;
  int nsoc = 0;                   // SOC table
  int *soc_off = nullptr, *soc_dim = nullptr, *soc_branch = nullptr;
  std::vector<int> soc_cone_index;
  int ncone3 = 0;                 // exponential / power cones (cone3.hip)
  int *c3_off = nullptr, *c3_kind = nullptr, *c3_branch = nullptr;
  real* c3_alpha = nullptr;
  std::vector<int> c3_cone_index;
  std::vector<CustomCone> custom;  // user-defined cones (custom.hip)
  real* custom_host = nullptr;   // pinned staging, sum of the custom dims
  int* custom_halt = nullptr;      // pinned copy of ctl->halt
  PsdPlan* psd = nullptr;
  void* psd_polar = nullptr;      // PolarPlan (psd_polar.hip): large cones
  void* accel = nullptr;          // AaState (anderson.hip)
  long long safeguarding_iter = 0;
  bool cg_sr = false;             // single-reduction (Chronopoulos-Gear) CG, cg_sr.hip: kkt_kind COSMO_HIP_KKT_CG_SR (or the lab switch behind cg_sr_auto)
  bool cg_sr_auto = false;        // lab switch COSMO_HIP_CG_SR_DEFAULT=1: kkt_kind CG took the one-launch recurrence on an assembled operator (measured and rejected as the default, api.hip: choose_cg_recurrence)
  long long auto_rho_fixed_at = -1;   // iteration at which the automatic rho interval (adaptive_rho_interval == 0, solver.jl:244-256) was fixed; -1: not (yet)
  bool cg_jacobi = false;  // kkt_kind CG_JACOBI: opt-in Jacobi-preconditioned CG on the assembled operator (cg_fold.hip)
  void* sr_rec = nullptr;         // 2 n records {r, w, s, p}
  real* cg_ru = nullptr;        // {r_i, u_i} interleaved (2n doubles): operands of the fused direction + A-product kernel (k_cg_dirA); null = unfused
  // persistent single-launch CG (cg_persist.hip)
  bool pcg_on = false;
  unsigned* pcg_sync = nullptr;
  real* pcg_u2 = nullptr;
  int pcg_W = 0, pcg_cap = 0;
  size_t pcg_smem = 0;
  long long pcg_launches = 0, pcg_fallbacks = 0;
  // clique sharding (comm.hip): this rank projects the SOC / PSD cones cone_lo <= k < cone_hi (cone_hi < 0: all cones)
  void* comm = nullptr;
  long long cone_lo = 0, cone_hi = -1;
  // row sharding (rowshard.hip, SURVEY 8e option 2 on a replicated CG): this rank owns the rows row_lo <= i < row_lo + m of the
  // m_g-row problem -- its cones, its rows of A (h->A), its columns of A' (h->AT), its slices of b / rho / Einv / s / mu / w_s.
  // Everything of length n is replicated and bit-identical on all ranks.  What stays GLOBAL: the reduced operator of the CG
  // (Am / PTm / fold, built from the whole A before the conversion) and the vectors it is refreshed from (rho_g, rho_cls_g).
  bool row_shard = false;
  long long m_g = 0, row_lo = 0;
  real* rho_g = nullptr;            // rho over all m_g rows (identical on every rank: a function of ctl->rho and the row classes)
  int* rho_cls_g = nullptr;
  real* red_n = nullptr;            // n + 2 nranks: partial A' products awaiting the all-reduce, then per-rank residual norms
  ConeTable cones_g;                // the whole composite set (h->cones = this rank's cones, offsets relative to row_lo)
  int n_bb = 0;                     // partials of ||rhs||^2 that the solve start folds (set by enqueue_cg_start)
  long long rs_allreduces = 0, rs_allreduce_elems = 0;
  // loop state
  real *w = nullptr, *w_prev = nullptr, *s = nullptr, *mu = nullptr, *s_tl = nullptr;
  real *ls_x = nullptr, *ls_s = nullptr, *x_tl = nullptr, *nu = nullptr;
  real *rhs = nullptr, *r = nullptr, *u = nullptr, *c = nullptr, *tmp_m = nullptr, *y2 = nullptr;
  real *mr = nullptr;           // MINRES work vectors
  real* partials = nullptr;     // COSMO_NSLOTS x COSMO_MAX_PARTIALS
  Ctl* ctl = nullptr;
  Ctl* ctl_host = nullptr;        // pinned mirror
  real* io = nullptr;           // staging buffer, n+m
  // parameters
  cosmo_hip_params prm;
  // host-side bookkeeping of the speculative enqueue
  long long host_iter = 0;        // ADMM iterations enqueued
  long long host_solves = 0;      // KKT solves enqueued
  int budget = 12;                // Krylov iterations enqueued per solve
  long long stalls = 0;
  // Krylov budget feedback (api.hip: solve_budget): Krylov count of every solve of the loop, copied back asynchronously
  static const int FB_RING = 64;
  int* fb_k = nullptr;                 // pinned ring, entry s % FB_RING = cg_k of loop solve s
  hipEvent_t fb_ev[FB_RING] = {};      // recorded behind the copy of solve s
  long long fb_recorded = 0;           // loop solves recorded so far
  long long fb_from = 0;               // first solve whose count describes the CURRENT regime (start, rho change or stall)
  int fb_mode = 1;                     // COSMO_HIP_BUDGET_FEEDBACK=0: window maximum + 2 only
  long long fb_stalls = 0;             // stalls of solves that ran on a feedback budget (three of them switch it off)
  bool fb_used[FB_RING] = {};          // entry s % FB_RING: loop solve s (1-based, as ctl->solves + 1) was enqueued with a feedback budget
  int cg_k_likely = 0x7fffffff;        // Krylov iterations from this index on are expected to be no-ops (set per solve by solve_budget)
  long long spmv_calls[3] = {0, 0, 0};
  // profiling
  bool profiling = false;      // HIP events around every loop kernel
  bool exact_launches = false; // synchronise after every Krylov iteration (no guarded no-op launches)
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, int>> ev_open;  // (class, event index of start)
  size_t ev_used = 0;
  double kc_seconds[COSMO_HIP_NUM_KERNEL_CLASSES] = {0};
  long long kc_launches[COSMO_HIP_NUM_KERNEL_CLASSES] = {0};
  hipEvent_t ev_proj0 = nullptr, ev_proj1 = nullptr;
};

// ---- error handling ------------------------------------------------------------------------------------------------
int32_t cosmo_fail(cosmo_hip_handle* h, int32_t code, const char* fmt, ...);
#define HIPCHK(h, call)                                                                                  \
  do {                                                                                                   \
    hipError_t e__ = (call);                                                                             \
    if (e__ != hipSuccess)                                                                               \
      return cosmo_fail((h), COSMO_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                             \
  } while (0)
#define CHK(call)                         \
  do {                                    \
    int32_t rc__ = (call);                \
    if (rc__ != COSMO_HIP_OK) return rc__; \
  } while (0)

// ---- launch helpers implemented in kernels.hip ---------------------------------------------------------------------
void build_row_blocks(const std::vector<int>& rowptr, int nrows, std::vector<int>& rb, int tile_override);   // api.hip: row boundaries of the CSR-stream tiles
// tile_override > 0: nonzeros per CSR-stream tile (<= COSMO_NNZ_PER_BLOCK) instead of the size heuristic of build_row_blocks
int32_t upload_csr(cosmo_hip_handle* h, const HostCsr& M, CsrDev& D, int split_col, int tile_override = 0);
void free_csr(CsrDev& D);
void prof_begin(cosmo_hip_handle* h, int kc);
void prof_end(cosmo_hip_handle* h);
int32_t prof_collect(cosmo_hip_handle* h);

// CG operator split
int32_t build_op_split(cosmo_hip_handle* h, bool force = false);
int32_t choose_cg_recurrence(cosmo_hip_handle* h);      // api.hip: kkt_kind CG on an assembled operator -> the one-launch recurrence of cg_sr.hip
int32_t refresh_op_split(cosmo_hip_handle* h);
void free_op_split(cosmo_hip_handle* h);

// assembled reduced operator (cg_fold.hip)
int32_t fold_build(cosmo_hip_handle* h, const HostCsr& Am, const std::vector<int>& prp, const std::vector<int>& pcol,
                   const std::vector<real>& pval);
int32_t fold_refresh(cosmo_hip_handle* h);
void fold_free(cosmo_hip_handle* h);
int32_t fold_enqueue_start(cosmo_hip_handle* h, int guard, real tol_k);
int32_t fold_enqueue_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count);

// plain y = M x (fine-grained ABI + building block)
int32_t launch_spmv_plain(cosmo_hip_handle* h, const CsrDev& M, const real* x, real* y);

// loop pieces (loop.hip)
int32_t enqueue_projection(cosmo_hip_handle* h, const real* src, real* dst, real* w_prev_dst,
                           const real* w_src, bool in_loop);
int32_t enqueue_admm_x_and_w(cosmo_hip_handle* h);
int32_t sync_ctl(cosmo_hip_handle* h);

int32_t comm_allreduce_flag(cosmo_hip_handle* h, int* flag);   // comm.hip: max over the ranks of a 0/1 flag

// single-reduction CG (cg_sr.hip)
int32_t sr_alloc(cosmo_hip_handle* h);
void sr_free(cosmo_hip_handle* h);
int32_t sr_enqueue_start(cosmo_hip_handle* h, int guard);
int32_t sr_enqueue_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count);

// merged launches over one-problem batches of different structure (batch.hip; used by batch_group.hip)
bool batch_multi_supported(const cosmo_hip_batch* b);
bool batch_multi_needs_ext(const cosmo_hip_batch* b);
int32_t batch_multi_optimize(cosmo_hip_batch** bs, int count, bool ext, cosmo_hip_result* results);

// persistent CG (cg_persist.hip)
int32_t pcg_setup(cosmo_hip_handle* h);
void pcg_free(cosmo_hip_handle* h);
int32_t pcg_enqueue_solve(cosmo_hip_handle* h, int guard);

// Anderson acceleration (anderson.hip)
void aa_free(cosmo_hip_handle* h);
bool aa_enabled(const cosmo_hip_handle* h);
bool aa_safeguarded(const cosmo_hip_handle* h);
bool aa_active(const cosmo_hip_handle* h);
int32_t aa_restart(cosmo_hip_handle* h);
void aa_note_rho_restart(cosmo_hip_handle* h);
int32_t aa_begin_solve(cosmo_hip_handle* h);
int32_t aa_enqueue_pre(cosmo_hip_handle* h, long long it, bool* attempted);
int32_t aa_fetch_flags(cosmo_hip_handle* h, int* success, int* declined);
int32_t aa_enqueue_guard(cosmo_hip_handle* h);
int32_t aa_enqueue_reset(cosmo_hip_handle* h);
void aa_count(cosmo_hip_handle* h, int accelerated, int declined);
void aa_check_accuracy_activation(cosmo_hip_handle* h, real r_prim, real r_dual, real max_norm_prim, real max_norm_dual);

// exponential / power cones (cone3.hip)
int32_t cone3_plan_create(cosmo_hip_handle* h);
void cone3_free(cosmo_hip_handle* h);
int32_t cone3_enqueue_project(cosmo_hip_handle* h, real* s, int guard);
int32_t cone3_enqueue_in_dual_neg(cosmo_hip_handle* h, const real* v, real tol, int* flag);
int32_t cone3_get_branches(cosmo_hip_handle* h, int32_t* out_per_cone);

// user-defined cones (custom.hip)
int32_t custom_plan_create(cosmo_hip_handle* h);
void custom_free(cosmo_hip_handle* h);
int32_t custom_enqueue_project(cosmo_hip_handle* h, real* s, int guard);
// which 0: in_dual(-v) (v = normalised -dy), 1: in_pol_recc(v); *ok is and-ed with the verdict of every custom cone
int32_t custom_test(cosmo_hip_handle* h, const real* v_dev, int which, real tol, bool* ok);

// PSD projection of large cones by the matrix-sign iteration (psd_polar.hip)
int32_t polar_plan_create(cosmo_hip_handle* h);
void polar_plan_destroy(cosmo_hip_handle* h);
bool polar_enabled(const cosmo_hip_handle* h);
int32_t polar_enqueue_project(cosmo_hip_handle* h, real* s, int guard);
int32_t polar_enqueue_project_batch(cosmo_hip_handle* h, real* s, int guard);
int32_t polar_adapt(cosmo_hip_handle* h);   // host-side schedule adaptation at a synchronisation point
bool polar_has_batch(const cosmo_hip_handle* h);
bool polar_has_large(const cosmo_hip_handle* h);

// PSD projection (psd.hip)
int32_t psd_plan_create(cosmo_hip_handle* h);
void psd_plan_destroy(cosmo_hip_handle* h);
int32_t psd_enqueue_project(cosmo_hip_handle* h, real* s, bool guard);
int32_t psd_get_ranks(cosmo_hip_handle* h, int64_t* rank_per_cone);
static inline bool cone_owned(const cosmo_hip_handle* h, long long k) { return h->cone_hi < 0 || (k >= h->cone_lo && k < h->cone_hi); }
// infeas.hip
int32_t infeas_enqueue_capture(cosmo_hip_handle* h);
int32_t infeas_check(cosmo_hip_handle* h, int32_t* status);
// comm.hip
int32_t comm_enqueue_exchange(cosmo_hip_handle* h, real* s);
int32_t comm_allgather_rows(cosmo_hip_handle* h, real* full);                 // in-place all-gather of the ranks' row slices of a full-length vector
int32_t comm_allreduce_sum(cosmo_hip_handle* h, real* buf, size_t count);     // in place, stream-ordered; identical bits on every rank
int32_t comm_allreduce_host(cosmo_hip_handle* h, double* vals, int count, int op /*0 sum, 1 max*/);   // synchronous, host scalars
int comm_nranks(const cosmo_hip_handle* h);
int comm_rank(const cosmo_hip_handle* h);
// rowshard.hip
int32_t rs_enqueue_cg_rhs(cosmo_hip_handle* h, int guard);
int32_t rs_enqueue_check(cosmo_hip_handle* h, int guard, int mode);
int32_t rs_get_iterates(cosmo_hip_handle* h, real* w, real* w_prev, real* s, real* mu);
void rs_free(cosmo_hip_handle* h);
extern "C" int32_t cosmo_hip_comm_destroy(cosmo_hip_handle* h);
