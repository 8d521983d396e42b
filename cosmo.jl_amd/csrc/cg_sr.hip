// cg_sr.hip -- single-reduction conjugate gradients (Chronopoulos-Gear) for the reduced KKT system, an OPT-IN alternative
// (kkt_kind = COSMO_HIP_KKT_CG_SR) to the literal cg! recurrence of IterativeSolvers v0.9 that the default CG path restates
// (src/linear_solver/kktsolver_indirect.jl:57-70).  Same operator L = P + sigma I + A' rho A, same stopping rule (||r|| <= abstol checked
// BEFORE each iteration, maxiter = n), same warm start; algebraically the same iterates, but the two inner products of an iteration
// are taken in ONE place, which lets an iteration run as TWO launches instead of three or four:
//
//   standard CG                                   Chronopoulos-Gear (w = L r, s = L p kept by recurrence)
//     beta = g / g_old ; p = r + beta p             beta = g / g_old ; alpha = g / (d - beta g / alpha_old)      g = r'r, d = w'r
//     c = L p ; alpha = g / (p'c)                   p = r + beta p ; s = w + beta s
//     x += alpha p ; r -= alpha c ; g = r'r         x += alpha p ; r -= alpha s ; w = L r ; g = r'r ; d = w'r
//
// Launch 1 (k_sr_update_A): every workgroup folds the partials of g and d, applies the stopping rule, updates its slice of p, s, x, r
// and computes tmp = rho .* (A r_new) with r_new REBUILT at the gathered columns from the 32-byte record {r, w, s, p} of the column
// (one gather per nonzero; the records are double-buffered because owners write new records while others gather old ones).
// Launch 2 (k_sr_op): w = P r + (sigma r + A' tmp) [+ diag .* r] through the row-merged operator, partials of w'r.
// Rounding differs from the literal recurrence (s is a recurrence, not a product), so this solver is NOT bit-comparable with the
// default one: tests/test_gpu_cg_sr.py holds it to the KKT tolerance of SURVEY 8c (dense solve 1e-8 in tight mode, iteration counts
// within 3 %, ADMM trajectories within 1e-7 of the literal solver; measured 1e-13).  The bench states which solver ran.
// Measured (profiles/r02_cg_variants.md): BASELINE config 5 (split operator, latency-bound launches) 43.5 vs 49.1 us per Krylov
// iteration (-11 %); config 2 (gather-bound) 44.7 vs 41.6 us (+7 %: the 32-byte record gather of launch 1 costs more than the launch
// it saves) -- hence opt-in, and the literal recurrence stays the default everywhere.  On the ASSEMBLED operator of cg_fold.hip the
// whole iteration is ONE launch (k_sr_M below): config 5 146.7 (literal, two launches) -> 158.2 it/s (profiles/r02_cfg5_cg_variants.json).
#include "device_utils.h"

struct SrRec { real r, w, s, p; };   // 32 bytes, one per variable

#define PARTS(h, slot) ((h)->partials + (size_t)(slot) * COSMO_MAX_PARTIALS)

// gamma / delta partials -> scalars of iteration k; returns false when the solve is (or has just been found) finished
struct SrScalars { real res, alpha, beta; };
__device__ __forceinline__ bool sr_scalars(Ctl* __restrict__ ctl, int k, long long maxiter, real pg, real pd, real* red, SrScalars& o) {
  const real tol = ctl->tol;
  const real g_old = (k > 0) ? ctl->sr_gamma[(k - 1) & 1] : 1.0;
  const real a_old = (k > 0) ? ctl->sr_alpha[(k - 1) & 1] : 1.0;
  const real g = block_sum(pg, red);
  const real d = block_sum(pd, red);
  const real res = sqrt(g);
  const bool done = (k >= maxiter) || (res <= tol);
  o.res = res;
  o.beta = (k > 0) ? g / g_old : 0.0;
  const real den = (k > 0) ? d - o.beta * g / a_old : d;
  o.alpha = g / den;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) ctl->cg_done = 1;
    ctl->resv[k & 1] = res;
    if (!done) { ctl->sr_gamma[k & 1] = g; ctl->sr_alpha[k & 1] = o.alpha; ctl->cg_k = k + 1; }
  }
  return !done;
}

// check_only != 0: evaluate the stopping rule after the last budgeted iteration (one workgroup)
__global__ __launch_bounds__(COSMO_BS) void k_sr_update_A(Ctl* __restrict__ ctl, int guard, int k, int check_only, long long n, long long maxiter,
                                                          const real* __restrict__ part_g_in, int n_g, const real* __restrict__ part_d, int n_d,
                                                          real* __restrict__ part_g_out, CsrView A, const SrRec* __restrict__ old_rec,
                                                          SrRec* __restrict__ new_rec, real* __restrict__ x, real* __restrict__ rvec,
                                                          const real* __restrict__ rho, real* __restrict__ tmp) {
  const real pg = partials_prefetch_sum(part_g_in, n_g);
  const real pd = partials_prefetch_sum(part_d, n_d);
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  SrScalars sc;
  if (check_only) {
    const real g = block_sum(pg, red);
    const real res = sqrt(g);
    if (threadIdx.x == 0 && ((k >= maxiter) || (res <= ctl->tol))) { ctl->cg_done = 1; ctl->resv[k & 1] = res; }
    return;
  }
  if (!sr_scalars(ctl, k, maxiter, pg, pd, red, sc)) return;
  const real alpha = sc.alpha, beta = sc.beta;
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const SrRec v = old_rec[i];
    const real p = v.r + beta * v.p;
    const real s = v.w + beta * v.s;
    x[i] = x[i] + alpha * p;
    const real rn = v.r - alpha * s;
    SrRec o; o.r = rn; o.w = 0.0; o.s = s; o.p = p;      // w is filled by k_sr_op
    new_rec[i] = o;
    rvec[i] = rn;                                        // compact copy: the operator kernel gathers 8-byte r, not 32-byte records
    acc += rn * rn;
  }
  for (int b = blockIdx.x; b < A.nb; b += gridDim.x) {
    const int4 d = reinterpret_cast<const int4*>(A.rb)[tile_of_block(b, A.nb, A.xcd_affine)];
    csr_stream_rows_g(A, [&](int c) { const SrRec v = old_rec[c]; return v.r - alpha * (v.w + beta * v.s); }, d.x, d.y, d.z, d.w, lds, red,
                      [&](int r, real s1, real s2) { tmp[r] = (s1 + s2) * rho[r]; });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_g_out[blockIdx.x] = acc;
}

// w = P r + (sigma r + A' tmp) [+ diag .* r] ; partials of w'r.  r is gathered from the compact vector r0 (kept next to the records:
// a random 8-byte gather from 32-byte records would quadruple the footprint of the gathered data in the L2s).  init != 0 (solve
// start): the record {r, w, 0, 0} is created; otherwise w is written into the record of the current iterate.
__global__ __launch_bounds__(COSMO_BS) void k_sr_op(const Ctl* __restrict__ ctl, int guard, int init, CsrView PT, real sigma, const real* __restrict__ r0,
                                                    SrRec* __restrict__ rec, const real* __restrict__ tmp, const real* __restrict__ diag,
                                                    real* __restrict__ part_d) {
  if (guard && ctl->halt) return;
  if (!init && ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  const int split_col = PT.split_col;
  real acc = 0.0;
  const int first_tile = tile_of_block(blockIdx.x, PT.nb, PT.xcd_affine);
  for (int t = first_tile; t < PT.nb; t += gridDim.x) {
    const int4 d = reinterpret_cast<const int4*>(PT.rb)[t];
    csr_stream_rows_g(PT, [&](int c) { return (c < split_col) ? r0[c] : tmp[c - split_col]; }, d.x, d.y, d.z, d.w, lds, red,
                      [&](int row, real s1, real s2) {
                        const real rj = r0[row];
                        real wj = s1 + (sigma * rj + s2);
                        if (diag) wj += diag[row] * rj;
                        if (init) { SrRec o; o.r = rj; o.w = wj; o.s = 0.0; o.p = 0.0; rec[row] = o; }
                        else rec[row].w = wj;
                        acc += wj * rj;
                      });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_d[PT.xcd_affine ? first_tile : (int)blockIdx.x] = acc;
}

// out = rho .* (A v) on a plain vector (solve start: A r0)
__global__ __launch_bounds__(COSMO_BS) void k_sr_A_plain(const Ctl* __restrict__ ctl, int guard, CsrView A, const real* __restrict__ v,
                                                         const real* __restrict__ rho, real* __restrict__ out) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  for (int b = blockIdx.x; b < A.nb; b += gridDim.x)
    csr_stream_tile(A, v, v, tile_of_block(b, A.nb, A.xcd_affine), lds, red, [&](int r, real s1, real s2) { out[r] = (s1 + s2) * rho[r]; });
}

// ---------------------------------------------------------------------------------------------------------------------
// The same recurrence on the ASSEMBLED operator M (cg_fold.hip): ONE launch per Krylov iteration.  The workgroup that owns the rows
// of a tile also owns those elements of the vectors: it folds the gamma / delta partials, applies the stopping rule, and for each of
// its rows updates {p, s, x, r} from the row's old record, computes w_new = (M r_new)_row with r_new REBUILT at the gathered columns
// (r_j - alpha (w_j + beta s_j): 24 bytes of the column's old record), writes the new record and the partials of r'r and w'r of
// the NEXT iteration.  Records and partial slots alternate by iteration parity (owners write new ones while others gather old ones).
//
// Round 6 measured this form as a candidate DEFAULT of kkt_kind CG on assembled operators (the reference's cg! lives in a package outside its
// tree, kktsolver_indirect.jl:66-74, and SURVEY 8(c) defines KKT parity as the residual bound) and rejected it: see api.hip: choose_cg_recurrence.
// The kernel has the load-first structure of k_cg_dirM: everything that does not depend on (alpha, beta) -- both partial sets, the tile
// descriptor, (col, val) and the 24-byte gathers of the first tile, the row pointers, the thread's own record and x -- is requested before
// the scalar work; the iteration index of a launch inside a captured chain comes from ctl->sr_k[parity].
// ---------------------------------------------------------------------------------------------------------------------
struct SrGat { real2 rw; real s; };     // what a gathered column contributes: {r, w} (one 16-byte load) and s
template <int SL>
__global__ __launch_bounds__(COSMO_BS) void k_sr_M(Ctl* __restrict__ ctl, int guard, int check_first, int k, int par, int check_only, long long maxiter,
                                                   const real* __restrict__ part_g_in, const real* __restrict__ part_d_in, int n_parts,
                                                   real* __restrict__ part_g_out, real* __restrict__ part_d_out, CsrView M,
                                                   const SrRec* __restrict__ old_rec, SrRec* __restrict__ new_rec, real* __restrict__ x) {
  if (check_first) { if (guard && ctl->halt) return; if (ctl->cg_done) return; }   // expected no-op (see k_cg_dirA): flags before any request
  const real pg = partials_prefetch_sum(part_g_in, n_parts);
  real pd = R(0.0);
  if (!check_only) pd = partials_prefetch_sum(part_d_in, n_parts);
  __shared__ real lds[SL * COSMO_BS > COSMO_NNZ_PER_BLOCK ? SL * COSMO_BS : COSMO_NNZ_PER_BLOCK];
  __shared__ real red[2 * (COSMO_BS / 64)];
  if (check_only) {               // stopping rule after the last budgeted iteration (one workgroup)
    if (guard && ctl->halt) return;
    if (ctl->cg_done) return;
    const real g = block_sum(pg, red);
    const real res = sqrt(g);
    if (threadIdx.x == 0 && ((k >= maxiter) || (res <= ctl->tol))) { ctl->cg_done = 1; ctl->resv[k & 1] = res; }
    return;
  }
  const int first_tile = tile_of_block(blockIdx.x, M.nb, M.xcd_affine);
  const bool have_tile = first_tile < M.nb;
  int4 d = make_int4(0, 0, 0, 0);
  if (have_tile) d = reinterpret_cast<const int4*>(M.rb)[first_tile];
  const int cnt0 = d.w - d.z;
  const bool fast = have_tile && cnt0 <= SL * COSMO_BS;       // a single long row takes the generic chunked path below
  real av[SL]; SrGat gv[SL];
  const real2* __restrict__ rec2 = reinterpret_cast<const real2*>(old_rec);
#pragma unroll
  for (int it = 0; it < SL; ++it) {
    const int kk = it * COSMO_BS + threadIdx.x;
    const bool ok = fast && kk < cnt0;
    const int e = ok ? d.z + kk : 0;
    const int cc = M.col[e];
    const real a = M.val[e];
    av[it] = ok ? a : R(0.0);
    gv[it].rw = rec2[2 * (size_t)cc];
    gv[it].s = old_rec[cc].s;
  }
  const int rfirst = d.x + threadIdx.x;
  const bool rowok = fast && rfirst < d.y;
  const int rr_ = rowok ? rfirst : 0;
  const int pa_ = M.rowptr[rr_], pb_ = M.rowptr[rr_ + 1];
  const real2 own_rw = rec2[2 * (size_t)rr_];
  const real2 own_sp = rec2[2 * (size_t)rr_ + 1];
  const real own_x = x[rr_];
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  if (k < 0) k = ctl->sr_k[par];                 // device-side index (captured chain, k >= 1): written by the launch in front, by nobody during this one
  const real tol = ctl->tol;
  const real g_old = (k > 0) ? ctl->sr_gamma[par ^ 1] : R(1.0);
  const real a_old = (k > 0) ? ctl->sr_alpha[par ^ 1] : R(1.0);
  real g = pg, dd = pd;
  block_sum2(g, dd, red);
  const real res = sqrt(g);
  const bool done = (k >= maxiter) || (res <= tol);
  const real beta = (k > 0) ? g / g_old : R(0.0);
  const real den = (k > 0) ? dd - beta * g / a_old : dd;
  const real alpha = g / den;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) ctl->cg_done = 1;
    ctl->resv[par] = res;
    if (!done) { ctl->sr_gamma[par] = g; ctl->sr_alpha[par] = alpha; ctl->cg_k = k + 1; ctl->sr_k[par ^ 1] = k + 1; }
  }
  if (done) return;
  real accg = R(0.0), accd = R(0.0);
  auto row_update = [&](int row, const real2 rw, const real2 sp, real xv, real wn) {
    const real p = rw.x + beta * sp.y;
    const real s = rw.y + beta * sp.x;
    x[row] = xv + alpha * p;
    const real rn = rw.x - alpha * s;
    SrRec o; o.r = rn; o.w = wn; o.s = s; o.p = p;
    new_rec[row] = o;
    accg += rn * rn;
    accd += wn * rn;
  };
  if (fast) {
#pragma unroll
    for (int it = 0; it < SL; ++it) {
      const int kk = it * COSMO_BS + threadIdx.x;
      if (kk < cnt0) lds[kk] = av[it] * (gv[it].rw.x - alpha * (gv[it].rw.y + beta * gv[it].s));
    }
    __syncthreads();
    if (rowok) row_update(rfirst, own_rw, own_sp, own_x, lds_seq_sum(lds, pa_ - d.z, pb_ - d.z));     // first row of this thread: pointers, record and x already here
    for (int r = rfirst + COSMO_BS; r < d.y; r += COSMO_BS) {
      const real wn = lds_seq_sum(lds, M.rowptr[r] - d.z, M.rowptr[r + 1] - d.z);
      row_update(r, rec2[2 * (size_t)r], rec2[2 * (size_t)r + 1], x[r], wn);
    }
    __syncthreads();
  }
  for (int t = fast ? first_tile + (int)gridDim.x : first_tile; t < M.nb; t += gridDim.x) {
    const int4 e = reinterpret_cast<const int4*>(M.rb)[t];
    csr_stream_rows_g(M, [&](int c) { const SrRec v = old_rec[c]; return v.r - alpha * (v.w + beta * v.s); }, e.x, e.y, e.z, e.w, lds, red,
                      [&](int row, real s1, real s2) { row_update(row, rec2[2 * (size_t)row], rec2[2 * (size_t)row + 1], x[row], s1 + s2); });
  }
  block_sum2(accg, accd, red);
  if (threadIdx.x == 0) {
    const int slot = M.xcd_affine ? first_tile : (int)blockIdx.x;
    part_g_out[slot] = accg; part_d_out[slot] = accd;
  }
}

// solve start on the assembled operator: records {r0, w0 = M r0, 0, 0}, partials of w0'r0 (r0 and its r'r partials come from k_fold_start)
__global__ __launch_bounds__(COSMO_BS) void k_sr_M_init(const Ctl* __restrict__ ctl, int guard, CsrView M, const real* __restrict__ r0,
                                                        SrRec* __restrict__ rec, real* __restrict__ part_d) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real acc = R(0.0);
  const int first_tile = tile_of_block(blockIdx.x, M.nb, M.xcd_affine);
  for (int t = first_tile; t < M.nb; t += gridDim.x) {
    csr_stream_tile(M, r0, r0, t, lds, red, [&](int row, real s1, real s2) {
      const real rj = r0[row], wj = s1 + s2;
      SrRec o; o.r = rj; o.w = wj; o.s = R(0.0); o.p = R(0.0);
      rec[row] = o;
      acc += wj * rj;
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_d[M.xcd_affine ? first_tile : (int)blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
int32_t sr_alloc(cosmo_hip_handle* h) {
  if (h->sr_rec) { (void)hipFree(h->sr_rec); h->sr_rec = nullptr; }
  if (!h->cg_sr) return COSMO_HIP_OK;
  const size_t bytes = 2 * sizeof(SrRec) * (size_t)std::max<long long>(h->n, 1);
  HIPCHK(h, hipMalloc((void**)&h->sr_rec, bytes));
  HIPCHK(h, hipMemsetAsync(h->sr_rec, 0, bytes, h->stream));
  return COSMO_HIP_OK;
}
void sr_free(cosmo_hip_handle* h) { if (h->sr_rec) { (void)hipFree(h->sr_rec); h->sr_rec = nullptr; } }

static inline int sr_grid1(const cosmo_hip_handle* h, const CsrDev& Ao) {
  long long ge = (h->n + COSMO_BS - 1) / COSMO_BS;
  if (ge > 1024) ge = 1024;
  return (int)std::max<long long>(std::max(Ao.grid, 1), ge);
}

// after enqueue_cg_start (r0 = rhs - L x0 in h->r, its r'r partials in SLOT_RR, tolerance set): w0 = L r0, records, w0'r0 partials
int32_t sr_enqueue_start(cosmo_hip_handle* h, int guard) {
  if (h->op_fold) {                     // assembled operator: one product (k_fold_start left r0 in h->r, its r'r partials in SLOT_RR)
    FoldPlan* f = (FoldPlan*)h->fold;
    prof_begin(h, KC_OP_APPLY);
    hipLaunchKernelGGL(k_sr_M_init, dim3(f->M.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(f->M), (const real*)h->r, (SrRec*)h->sr_rec,
                       PARTS(h, SLOT_UC));
    prof_end(h);
    h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
    HIPCHK(h, hipGetLastError());
    return COSMO_HIP_OK;
  }
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  const real* rho_o = h->op_split ? h->op_rho_m : h->rho;
  const real* diag_o = h->op_split ? h->op_diag : nullptr;
  SrRec* rec0 = (SrRec*)h->sr_rec;
  prof_begin(h, KC_SPMV_A);
  hipLaunchKernelGGL(k_sr_A_plain, dim3(std::max(Ao.grid, 1)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(Ao), h->r, rho_o, h->tmp_m);
  prof_end(h);
  prof_begin(h, KC_OP_APPLY);
  hipLaunchKernelGGL(k_sr_op, dim3(PTo.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 1, view_of(PTo), h->prm.sigma, h->r, rec0, h->tmp_m, diag_o,
                     PARTS(h, SLOT_UC));
  prof_end(h);
  h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// ---- assembled operator: launches of k_sr_M ---------------------------------------------------------------------------------------------
// k_arg >= 0: the iteration index as an argument (direct launches); k_arg < 0: read on the device from ctl->sr_k[par] (captured chain)
static void sr_launch_M(cosmo_hip_handle* h, FoldPlan* f, int guard, int check_first, int k_arg, int par) {
  SrRec* rec = (SrRec*)h->sr_rec;
  const long long n = h->n;
  const int G = f->M.grid;
  real* g_in = par ? PARTS(h, SLOT_AUX0) : PARTS(h, SLOT_RR);        // r'r partials by iteration parity
  real* d_in = par ? PARTS(h, SLOT_AUX1) : PARTS(h, SLOT_UC);        // w'r partials
  real* g_out = par ? PARTS(h, SLOT_RR) : PARTS(h, SLOT_AUX0);
  real* d_out = par ? PARTS(h, SLOT_UC) : PARTS(h, SLOT_AUX1);
  prof_begin(h, KC_OP_APPLY);
#define LAUNCH_SRM(SLN) hipLaunchKernelGGL((k_sr_M<SLN>), dim3(G), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, check_first, k_arg, par, 0, n, (const real*)g_in, \
                          (const real*)d_in, G, g_out, d_out, view_of(f->M), (const SrRec*)(rec + (size_t)par * n), rec + (size_t)(par ^ 1) * n, h->x_tl)
  switch (f->slots) {
    case 1: LAUNCH_SRM(1); break;
    case 2: LAUNCH_SRM(2); break;
    case 3: LAUNCH_SRM(3); break;
    case 4: LAUNCH_SRM(4); break;
    default: LAUNCH_SRM(8); break;
  }
#undef LAUNCH_SRM
  prof_end(h);
}

// Captured chain of speculative iterations, as cg_fold.hip's (same switches: COSMO_HIP_CG_GRAPH, COSMO_HIP_CG_GRAPH_LEN).  The chain starts at an
// ODD iteration (k = 1 is the first one behind the direct k = 0 launch) and has an EVEN number of iterations, so that every launch of it sees the
// records / partial slots of its parity whatever the number of chains in front.
static bool sr_chain_ready(cosmo_hip_handle* h, FoldPlan* f) {
  if (f->chain_off || h->profiling) return false;
  if (f->sr_chain) return true;
  if (const char* e = getenv("COSMO_HIP_CG_GRAPH")) { if (atoi(e) == 0) { f->chain_off = 1; return false; } }
  int len = 16;
  if (const char* e = getenv("COSMO_HIP_CG_GRAPH_LEN")) { const int v = atoi(e); if (v >= 1 && v <= 256) len = v; }
  len += len & 1;
  hipGraphExec_t ex[2] = {nullptr, nullptr};
  for (int cf = 0; cf < 2; ++cf) {               // the chain, and the chain of iterations that are expected to be no-ops
    hipGraph_t g = nullptr;
    bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      for (int i = 0; i < len; ++i) sr_launch_M(h, f, 1, cf, -1, (i + 1) & 1);
      ok = hipStreamEndCapture(h->stream, &g) == hipSuccess && g;
    }
    if (ok) ok = hipGraphInstantiate(&ex[cf], g, nullptr, nullptr, 0) == hipSuccess;
    if (g) (void)hipGraphDestroy(g);
    if (!ok) {
      (void)hipGetLastError();
      if (ex[0]) (void)hipGraphExecDestroy(ex[0]);
      f->chain_off = 1;
      return false;
    }
  }
  f->sr_chain = ex[0]; f->sr_chain_cf = ex[1]; f->sr_chain_len = len;
  return true;
}

int32_t sr_enqueue_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count) {
  const long long n = h->n;
  if (h->op_fold) {                     // ONE launch per Krylov iteration on the assembled operator
    FoldPlan* f = (FoldPlan*)h->fold;
    SrRec* rec = (SrRec*)h->sr_rec;
    const int G = f->M.grid;
    int k = k_begin;
    const int k_end = k_begin + count;
    if (guard == 1 && k_begin == 0 && count > 1 && sr_chain_ready(h, f)) {
      sr_launch_M(h, f, guard, 0, 0, 0);                         // k = 0: index as an argument (also initialises the device-side index for the chain)
      h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
      for (k = 1; k + f->sr_chain_len <= k_end; k += f->sr_chain_len) {   // whole chains; the remainder goes out directly below
        HIPCHK(h, hipGraphLaunch((hipGraphExec_t)(k >= h->cg_k_likely ? f->sr_chain_cf : f->sr_chain), h->stream));
        h->spmv_calls[0] += f->sr_chain_len; h->spmv_calls[1] += f->sr_chain_len; h->spmv_calls[2] += f->sr_chain_len;
      }
    }
    for (; k < k_end; ++k) {
      sr_launch_M(h, f, guard, (k >= h->cg_k_likely) ? 1 : 0, k, k & 1);
      h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
    }
    const int kk = k_end, par = kk & 1;
    prof_begin(h, KC_CG_DIR);
    hipLaunchKernelGGL((k_sr_M<1>), dim3(1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 0, kk, par, 1, n, (const real*)(par ? PARTS(h, SLOT_AUX0) : PARTS(h, SLOT_RR)),
                       (const real*)nullptr, G, (real*)nullptr, (real*)nullptr, view_of(f->M), (const SrRec*)rec, rec, h->x_tl);
    prof_end(h);
    HIPCHK(h, hipGetLastError());
    return COSMO_HIP_OK;
  }
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  const real* rho_o = h->op_split ? h->op_rho_m : h->rho;
  const real* diag_o = h->op_split ? h->op_diag : nullptr;
  SrRec* rec = (SrRec*)h->sr_rec;
  const int g1 = sr_grid1(h, Ao);
  auto gslot = [&](int k) { return (k & 1) ? PARTS(h, SLOT_AUX0) : PARTS(h, SLOT_RR); };
  auto gcount = [&](int k) { return (k == 0) ? PTo.grid : g1; };
  for (int k = k_begin; k < k_begin + count; ++k) {
    SrRec* old_rec = rec + (size_t)(k & 1) * n;
    SrRec* new_rec = rec + (size_t)((k + 1) & 1) * n;
    prof_begin(h, KC_SPMV_A);
    hipLaunchKernelGGL(k_sr_update_A, dim3(g1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, 0, n, n, gslot(k), gcount(k), PARTS(h, SLOT_UC), PTo.grid,
                       gslot(k + 1), view_of(Ao), old_rec, new_rec, h->x_tl, h->r, rho_o, h->tmp_m);
    prof_end(h);
    prof_begin(h, KC_OP_APPLY);
    hipLaunchKernelGGL(k_sr_op, dim3(PTo.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 0, view_of(PTo), h->prm.sigma, (const real*)h->r, new_rec,
                       h->tmp_m, diag_o, PARTS(h, SLOT_UC));
    prof_end(h);
    h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  }
  const int kk = k_begin + count;
  prof_begin(h, KC_CG_DIR);
  hipLaunchKernelGGL(k_sr_update_A, dim3(1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, kk, 1, n, n, gslot(kk), gcount(kk), PARTS(h, SLOT_UC), PTo.grid,
                     gslot(kk + 1), view_of(Ao), rec, rec, h->x_tl, h->r, rho_o, h->tmp_m);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}
