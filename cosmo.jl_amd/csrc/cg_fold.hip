// cg_fold.hip -- the reduced KKT operator of the CG solve ASSEMBLED as one sparse matrix, for problems where that is cheap.
//
// reduced_mul! (src/linear_solver/kktsolver_indirect.jl:57-64) applies  x -> P x + sigma x + A'(rho .* (A x))  as two dependent sparse
// products, so a Krylov iteration of the device CG is at least three launches: [direction + A product] -> [P | A'] product + u'c ->
// [x, r update + r'r].  On a decomposed SDP (BASELINE config 5) each of those launches is a 4-6 us chain of dependent memory round
// trips over a ~100 k-nonzero operator, and ~170 such iterations per ADMM iteration are half of the whole run
// (profiles/r02_cfg5_timeline_before_fold.txt).  After the operator split (api.hip: build_op_split) only the rows of A with two or
// more nonzeros are left as a matrix Am; when Am' rho Am is sparse enough the operator is assembled here once,
//
//       M(rho) = P + diag(sigma + d(rho)) + Am' diag(rho_m) Am ,        d = the diagonal the single-nonzero rows contribute,
//
// and a Krylov iteration becomes TWO launches: k_cg_dirM (stopping rule, beta, u = r + beta u rebuilt at the gathered columns,
// c = M u, partials of u'c) and the unchanged k_cg_upd.  The solve start is one launch instead of two.
//
// Parity: the same linear operator with a different association (like the operator split itself): M's entries are sums of
// rho_k a_ki a_kj in a fixed order (Am rows ascending), row sums of M u run left to right over the columns.  The CG recurrence, the
// stopping rule and the tolerance are the literal cg! ones; iteration counts agree with the unfolded operator to +-1 per solve and
// trajectories to the tolerances of SURVEY 8c (tests/test_gpu_cg_fold.py).  rho may change on the device (adaptive rho): the values
// are rebuilt by k_fold_refresh right after k_op_refresh, unconditionally, every time the adaptation rule has run.
// COSMO_HIP_OP_FOLD=0 disables the assembly.
#include "device_utils.h"
#include <algorithm>

#define PARTS(h, slot) ((h)->partials + (size_t)(slot) * COSMO_MAX_PARTIALS)


// launch helpers of kernels.hip (every launch stays next to its kernel)
int32_t launch_cg_upd(cosmo_hip_handle* h, int guard, int k, int n_uc);
int32_t launch_cg_dir_check(cosmo_hip_handle* h, int guard, int kk, int n_rr);
static inline int ew_grid(long long N) {
  long long g = (N + COSMO_BS - 1) / COSMO_BS;
  if (g < 1) g = 1;
  if (g > COSMO_MAX_PARTIALS) g = COSMO_MAX_PARTIALS;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------------
// values of M from rho: one thread per stored entry, terms added in their stored (Am-row ascending) order
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_fold_refresh(long long nnz, const real* __restrict__ base, const int* __restrict__ drow,
                                                           const int* __restrict__ tptr, const int* __restrict__ trow,
                                                           const real* __restrict__ tprod, const real* __restrict__ rho_m,
                                                           const real* __restrict__ diag, real sigma, real* __restrict__ val,
                                                           const int* __restrict__ ppos, real* __restrict__ pval) {
  for (long long p = (long long)blockIdx.x * COSMO_BS + threadIdx.x; p < nnz; p += (long long)gridDim.x * COSMO_BS) {
    real s = 0.0;
    for (int t = tptr[p]; t < tptr[p + 1]; ++t) s += rho_m[trow[t]] * tprod[t];
    const int i = drow[p];
    const real v = (i >= 0) ? base[p] + ((sigma + diag[i]) + s) : base[p] + s;
    val[p] = v;
    if (ppos) pval[ppos[p]] = v;                         // the tile-major padded copy k_cg_dirM streams
  }
}

// Jacobi preconditioner of the opt-in PCG: dinv_i = 1 / M_ii, read from the assembled values (dpos[i] = index of row i's diagonal entry,
// which fold_build always stores) right after every k_fold_refresh
__global__ __launch_bounds__(COSMO_BS) void k_fold_dinv(long long n, const int* __restrict__ dpos, const real* __restrict__ val, real* __restrict__ dinv) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) dinv[i] = R(1.0) / val[dpos[i]];
}

// ---------------------------------------------------------------------------------------------------------------------
// solve start: r = rhs - M x (x = warm start), {r, 0} records for the first direction, partials of r'r, abstol = tol_k / ||rhs||
// (kktsolver_indirect.jl:70 ; cg! computes the initial residual with one operator application)
// ---------------------------------------------------------------------------------------------------------------------
// PC (opt-in Jacobi-preconditioned CG, COSMO_HIP_KKT_CG_JACOBI; IterativeSolvers' PCGIterable with Pl = Diagonal(diag M)): the records carry
// {z_i, u_i} with z = dinv .* r instead of {r_i, u_i}, and a second set of partials holds z'r.
template <bool PC>
__global__ __launch_bounds__(COSMO_BS) void k_fold_start(Ctl* __restrict__ ctl, int guard, CsrView M, const real* __restrict__ x,
                                                         const real* __restrict__ rhs, real* __restrict__ r, real2* __restrict__ ru,
                                                         real* __restrict__ part_rr, const real* __restrict__ part_bb, int n_bb, real tol_k,
                                                         const real* __restrict__ dinv, real* __restrict__ part_rz, const real* __restrict__ tx) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  if (blockIdx.x == 0) {
    const real bb = reduce_partials_sum(part_bb, n_bb, red);
    if (threadIdx.x == 0) {
      const real nb = sqrt(bb);
      ctl->rhs_norm = nb;
      ctl->tol = tol_k / nb;
    }
  }
  real acc = 0.0, accz = 0.0;
  const int first_tile = tile_of_block(blockIdx.x, M.nb, M.xcd_affine);
  for (int k = first_tile; k < M.nb; k += gridDim.x) {
    csr_stream_tile(M, x, tx ? tx : x, k, lds, red, [&](int row, real s1, real s2) {      // (partial assembly: the columns >= split_col gather (Ad x) from tx)
      const real rj = rhs[row] - (s1 + s2);
      r[row] = rj;
      if constexpr (PC) {
        const real zj = dinv[row] * rj;                       // ldiv!(c, Pl, r)
        ru[row] = make_real2(zj, R(0.0));
        accz += zj * rj;
      } else {
        if (ru) ru[row] = make_real2(rj, R(0.0));
      }
      acc += rj * rj;
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_rr[M.xcd_affine ? first_tile : (int)blockIdx.x] = acc;
  if constexpr (PC) {
    accz = block_sum(accz, red);
    if (threadIdx.x == 0) part_rz[M.xcd_affine ? first_tile : (int)blockIdx.x] = accz;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Krylov step k, first half + the WHOLE operator (IterativeSolvers v0.9 cg.jl `iterate`): residual_k = ||r|| from the partials; stop if
// k >= maxiter or residual_k <= tol (checked BEFORE the iteration); beta = res_k^2 / res_{k-1}^2 (prev = 1 at k = 0, where the
// records carry u = 0); u_k = r + beta u_{k-1} rebuilt at the gathered columns from the 16-byte records {r_i, u_i}; c = M u_k;
// partials of u_k'c.  Same load-first structure as k_cg_dirA (kernels.hip): everything that does not depend on beta is requested
// before the scalar work.
// ---------------------------------------------------------------------------------------------------------------------
// SL = nonzero slots per thread of the register-staged first tile (tiles of the assembled matrix hold at most SL * 256 nonzeros; a
// longer single row takes the generic path).
// PC: beta = rho_k / rho_{k-1} with rho = z'r from the second partial set (ctl->sr_gamma, by iteration parity: the single-reduction CG that
// otherwise owns those two scalars is a different kkt_kind), u_k = z + beta u_{k-1} from the {z, u} records; the stopping rule stays ||r||_2.
template <int SL, bool PC>
__global__ __launch_bounds__(COSMO_BS) void k_cg_dirM(Ctl* __restrict__ ctl, int guard, int check_first, int k, long long n, long long maxiter,
                                                      const real* __restrict__ part_rr, int n_rr, CsrView M, const real2* __restrict__ ru,
                                                      real* __restrict__ c, real* __restrict__ u, real* __restrict__ part_uc,
                                                      const real* __restrict__ part_rz, const int* __restrict__ pcol, const real* __restrict__ pval,
                                                      const real2* __restrict__ tt, real* __restrict__ tcur, int nd) {
  // partial assembly (FoldPlan::nd > 0): a column cc >= n of the stored matrix [Ms | Ad' rho_d] gathers the record tt[cc - n] = {(Ad r)_kd, (Ad u_prev)_kd}; the
  // rebuilt value x + beta y is then (Ad u_k)_kd, just as {r, u} gives u_k for the columns < n.  nd == 0: every column is < n and REC is ru[cc].
#define REC(cc) (((cc) < (int)n) ? ru[(cc)] : tt[(cc) - (int)n])
  if (check_first) { if (guard && ctl->halt) return; if (ctl->cg_done) return; }   // expected no-op (see k_cg_dirA): flags before any request
  const real pa = partials_prefetch_sum(part_rr, n_rr);
  real pz = 0.0;
  if constexpr (PC) pz = partials_prefetch_sum(part_rz, n_rr);
  const int first_tile = tile_of_block(blockIdx.x, M.nb, M.xcd_affine);
  const bool have_tile = first_tile < M.nb;
  int4 d = make_int4(0, 0, 0, 0);
  if (have_tile) d = reinterpret_cast<const int4*>(M.rb)[first_tile];
  real av[SL]; real2 gv[SL];
  int cnt0;
  bool fast;
  if (pcol) {
    // tile-major padded copy (cap = SL * 256 slots per tile): the addresses depend on the block index only, so (col, val) are requested together
    // with the descriptor and the partials -- the dependent chain is col -> gather, one memory round trip shorter (the working set comes back
    // through the fabric at every kernel boundary: profiles/r04_cfg5_pmc_traffic.json).  Padding slots hold column 0 / value 0 and are masked by
    // cnt0 once the descriptor has arrived.
    int ccs[SL];
#pragma unroll
    for (int it = 0; it < SL; ++it) {
      const long long e = have_tile ? (long long)first_tile * (SL * COSMO_BS) + it * COSMO_BS + threadIdx.x : 0;
      ccs[it] = pcol[e];
      av[it] = pval[e];
    }
#pragma unroll
    for (int it = 0; it < SL; ++it) gv[it] = REC(ccs[it]);
    cnt0 = d.w - d.z;
    fast = have_tile && cnt0 <= SL * COSMO_BS;
#pragma unroll
    for (int it = 0; it < SL; ++it) if (!(fast && it * COSMO_BS + (int)threadIdx.x < cnt0)) av[it] = 0.0;
  } else {
    cnt0 = d.w - d.z;
    fast = have_tile && cnt0 <= SL * COSMO_BS;            // a single long row takes the generic chunked path below
#pragma unroll
    for (int it = 0; it < SL; ++it) {
      const int kk = it * COSMO_BS + threadIdx.x;
      const bool ok = fast && kk < cnt0;
      const int e = ok ? d.z + kk : 0;
      const int cc = M.col[e];
      const real a = M.val[e];
      av[it] = ok ? a : 0.0;
      gv[it] = REC(cc);
    }
  }
  const int rfirst = d.x + threadIdx.x;
  const bool rowok = fast && rfirst < d.y;
  const int rr_ = rowok ? rfirst : 0;
  const int pa_ = M.rowptr[rr_], pb_ = M.rowptr[rr_ + 1];
  const real2 rw = ru[rr_];                                       // {r, u_{k-1}} of this thread's row: u_k of the row for u'c
  const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
  const real2 own = ru[i0 < n ? i0 : 0];
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  if (k < 0) k = ctl->cg_k;                      // device-side index (captured chain, k >= 1): written by the k_cg_upd in front, by nobody during this kernel
  const real tol = ctl->tol;
  const real prev = (k == 0) ? 1.0 : ctl->resv[(k - 1) & 1];
  const real rr = block_sum(pa, red);
  const real res = sqrt(rr);
  const bool done = (k >= maxiter) || (res <= tol);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) ctl->cg_done = 1;
    ctl->resv[k & 1] = res;
    ctl->cg_kd = k;
  }
  if (done) return;
  real beta = (res * res) / (prev * prev);
  if constexpr (PC) {
    const real rz = block_sum(pz, red);
    const real rz_prev = (k == 0) ? R(1.0) : ctl->sr_gamma[(k - 1) & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->sr_gamma[k & 1] = rz;
    beta = rz / rz_prev;
  }
  if (i0 < n) u[i0] = own.x + beta * own.y;
  for (long long i = i0 + (long long)gridDim.x * COSMO_BS; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const real2 v = ru[i];
    u[i] = v.x + beta * v.y;
  }
  for (long long kd = i0; kd < nd; kd += (long long)gridDim.x * COSMO_BS) {      // (Ad u_k) for the record k_cg_updF writes: the same expression the gathers evaluate
    const real2 v = tt[kd];
    tcur[kd] = v.x + beta * v.y;
  }
  real acc = 0.0;
  if (fast) {
#pragma unroll
    for (int it = 0; it < SL; ++it) {
      const int kk = it * COSMO_BS + threadIdx.x;
      if (kk < cnt0) lds[kk] = av[it] * (gv[it].x + beta * gv[it].y);
    }
    __syncthreads();
    if (rowok) {                                    // first row of this thread: pointers and its own record already here
      const real cj = lds_seq_sum(lds, pa_ - d.z, pb_ - d.z);
      c[rfirst] = cj;
      acc += (rw.x + beta * rw.y) * cj;
    }
    for (int r = rfirst + COSMO_BS; r < d.y; r += COSMO_BS) {
      const real cj = lds_seq_sum(lds, M.rowptr[r] - d.z, M.rowptr[r + 1] - d.z);
      const real2 v = ru[r];
      c[r] = cj;
      acc += (v.x + beta * v.y) * cj;
    }
    __syncthreads();
  }
  for (int t = fast ? first_tile + (int)gridDim.x : first_tile; t < M.nb; t += gridDim.x) {
    const int4 dd = reinterpret_cast<const int4*>(M.rb)[t];
    csr_stream_rows_g(M, [&](int cc) { const real2 v = REC(cc); return v.x + beta * v.y; }, dd.x, dd.y, dd.z, dd.w, lds, red,
                      [&](int r, real s1, real s2) {
                        const real cj = s1 + s2;
                        const real2 v = ru[r];
                        c[r] = cj;
                        acc += (v.x + beta * v.y) * cj;
                      });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_uc[M.xcd_affine ? first_tile : (int)blockIdx.x] = acc;
#undef REC
}

// ---------------------------------------------------------------------------------------------------------------------
// Partial assembly: the dense rows of Am stay factored.  Small kernels on their CSR (one thread per dense row, left-to-right row sums):
//   k_ad_dot   : out[kd] = (Ad v)_kd            (solve start: Ad x for the start residual)
//   k_ad_rec   : tt[kd] = {(Ad r0)_kd, 0}        (solve start: the records of iteration 0)
//   k_cg_updF  : k_cg_upd<false> (x += alpha u ; r -= alpha c ; {r, u} records ; r'r partials) + for every dense row the record {(Ad r_new)_kd, (Ad u_k)_kd},
//                (Ad r_new) as a fresh CSR-stream product over the dense rows with r_new REBUILT at the gathered columns from the old (parity-buffered) records.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_ad_dot(const Ctl* __restrict__ ctl, int guard, CsrView Ad, const real* __restrict__ v, real* __restrict__ out,
                                                     real2* __restrict__ out_rec) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  for (int t = blockIdx.x; t < Ad.nb; t += gridDim.x)
    csr_stream_tile(Ad, v, v, t, lds, red, [&](int kd, real s1, real s2) { if (out_rec) out_rec[kd] = make_real2(s1 + s2, R(0.0)); else out[kd] = s1 + s2; });
}

// ADSL: nonzero slots per thread of a dense-row tile (tiles of Ad hold at most ADSL * 256 nonzeros; a longer single row takes the generic path)
#define ADSL 4
__global__ __launch_bounds__(COSMO_BS) void k_cg_updF(Ctl* __restrict__ ctl, int guard, int k, long long n, const real* __restrict__ part_uc, int n_uc,
                                                      const real* __restrict__ u, const real* __restrict__ c, real* __restrict__ x, real* __restrict__ r,
                                                      real* __restrict__ part_rr, const real2* __restrict__ ru_old, real2* __restrict__ ru_new, int gE, int nd,
                                                      CsrView Ad, real2* __restrict__ tt, const real* __restrict__ tcur) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  if ((int)blockIdx.x < gE) {
    // ---- workgroups [0, gE): the vector update of k_cg_upd<false>, r_old taken from the old records ------------------------------------------------
    const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
    real u0 = 0.0, c0 = 0.0, x0 = 0.0, r0 = 0.0;
    if (i0 < n) { u0 = u[i0]; c0 = c[i0]; x0 = x[i0]; r0 = ru_old[i0].x; }
    const real pa = partials_prefetch_sum(part_uc, n_uc);
    if (guard && ctl->halt) return;
    if (ctl->cg_done) return;
    if (k < 0) k = ctl->cg_kd;
    const real res = ctl->resv[k & 1];
    const real uc = block_sum(pa, red);
    const real alpha = (res * res) / uc;
    real acc = 0.0;
    if (i0 < n) {
      x[i0] = x0 + alpha * u0;
      const real ri = r0 - alpha * c0;
      r[i0] = ri;
      ru_new[i0] = make_real2(ri, u0);
      acc += ri * ri;
    }
    for (long long i = i0 + (long long)gE * COSMO_BS; i < n; i += (long long)gE * COSMO_BS) {
      const real ui = u[i];
      x[i] = x[i] + alpha * ui;
      const real ri = ru_old[i].x - alpha * c[i];
      r[i] = ri;
      ru_new[i] = make_real2(ri, ui);
      acc += ri * ri;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
      part_rr[blockIdx.x] = acc;
      if (blockIdx.x == 0) ctl->cg_k = k + 1;
    }
    return;
  }
  // ---- workgroups [gE, ...): tiles of the dense rows.  Load-first like k_cg_dirM: descriptor, (col, val), the gathers of r_j (old records) and c_j, the
  // row pointers and (Ad u_k) of the thread's row are all requested BEFORE the partial sums are folded -- none of them depends on alpha.  Absent slots read
  // a valid address and carry a zero value (a load under a per-thread `if` makes the compiler wait for it on the spot: four serial round trips here).
  const int first_tile = (int)blockIdx.x - gE;
  const bool have_tile = first_tile < Ad.nb;
  int4 d = make_int4(0, 0, 0, 0);
  if (have_tile) d = reinterpret_cast<const int4*>(Ad.rb)[first_tile];
  const int cnt0 = d.w - d.z;
  const bool fast = have_tile && cnt0 <= ADSL * COSMO_BS;
  real av[ADSL], gr[ADSL], gc[ADSL];
  int ccs[ADSL];
#pragma unroll
  for (int it = 0; it < ADSL; ++it) {
    const int kk = it * COSMO_BS + threadIdx.x;
    const bool ok = fast && kk < cnt0;
    const int e = ok ? d.z + kk : 0;
    ccs[it] = Ad.col[e];
    const real a = Ad.val[e];
    av[it] = ok ? a : R(0.0);
  }
#pragma unroll
  for (int it = 0; it < ADSL; ++it) { gr[it] = ru_old[ccs[it]].x; gc[it] = c[ccs[it]]; }
  const int rfirst = d.x + threadIdx.x;
  const bool rowok = fast && rfirst < d.y;
  const int rr_ = rowok ? rfirst : 0;
  const int pa_ = Ad.rowptr[rr_], pb_ = Ad.rowptr[rr_ + 1];
  const real tk_ = tcur[rr_];
  const real pa = partials_prefetch_sum(part_uc, n_uc);
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  if (k < 0) k = ctl->cg_kd;
  const real res = ctl->resv[k & 1];
  const real uc = block_sum(pa, red);
  const real alpha = (res * res) / uc;
  // (Ad r_new)_kd = sum a_kj (r_j - alpha c_j): r_j from the OLD records (stable during this launch), c complete; the owners evaluate the same expression.
  // (Measured and not kept: both row sums (Ad r), (Ad c) formed before alpha is known and combined behind it -- 11.8 instead of 11.3 us per Krylov iteration
  //  on BASELINE config 5, and +89 Krylov iterations over 61 tight solves: outside the +-1 per solve of the fold tests.)
  if (fast) {
#pragma unroll
    for (int it = 0; it < ADSL; ++it) {
      const int kk = it * COSMO_BS + threadIdx.x;
      if (kk < cnt0) lds[kk] = av[it] * (gr[it] - alpha * gc[it]);
    }
    __syncthreads();
    if (rowok) tt[rfirst] = make_real2(lds_seq_sum(lds, pa_ - d.z, pb_ - d.z), tk_);
    for (int kd = rfirst + COSMO_BS; kd < d.y; kd += COSMO_BS)
      tt[kd] = make_real2(lds_seq_sum(lds, Ad.rowptr[kd] - d.z, Ad.rowptr[kd + 1] - d.z), tcur[kd]);
    __syncthreads();
  }
  for (int t = fast ? first_tile + ((int)gridDim.x - gE) : first_tile; t < Ad.nb; t += (int)gridDim.x - gE) {
    const int4 dd = reinterpret_cast<const int4*>(Ad.rb)[t];
    csr_stream_rows_g(Ad, [&](int cc) { return ru_old[cc].x - alpha * c[cc]; }, dd.x, dd.y, dd.z, dd.w, lds, red,
                      [&](int kd, real s1, real s2) { tt[kd] = make_real2(s1 + s2, tcur[kd]); });
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
void fold_free(cosmo_hip_handle* h) {
  FoldPlan* f = (FoldPlan*)h->fold;
  h->op_fold = false;
  if (!f) return;
  free_csr(f->M);
  if (f->base) (void)hipFree(f->base);
  if (f->drow) (void)hipFree(f->drow);
  if (f->tptr) (void)hipFree(f->tptr);
  if (f->trow) (void)hipFree(f->trow);
  if (f->tprod) (void)hipFree(f->tprod);
  if (f->pcol) (void)hipFree(f->pcol);
  if (f->pval) (void)hipFree(f->pval);
  if (f->ppos) (void)hipFree(f->ppos);
  if (f->dpos) (void)hipFree(f->dpos);
  if (f->dinv) (void)hipFree(f->dinv);
  free_csr(f->Ad);
  if (f->tt) (void)hipFree(f->tt);
  if (f->tcur) (void)hipFree(f->tcur);
  if (f->tx) (void)hipFree(f->tx);
  if (f->chain) (void)hipGraphExecDestroy((hipGraphExec_t)f->chain);
  if (f->chain_cf) (void)hipGraphExecDestroy((hipGraphExec_t)f->chain_cf);
  if (f->sr_chain) (void)hipGraphExecDestroy((hipGraphExec_t)f->sr_chain);
  if (f->sr_chain_cf) (void)hipGraphExecDestroy((hipGraphExec_t)f->sr_chain_cf);
  delete f;
  h->fold = nullptr;
}

template <class T>
static int32_t up(cosmo_hip_handle* h, T** dst, const std::vector<T>& v) {
  HIPCHK(h, hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIPCHK(h, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return COSMO_HIP_OK;
}

// Am: the multi-nonzero rows of A (compact, columns ascending within a row); (prp, pcol, pval): CSR of P.  Called at the end of
// build_op_split (the split is active and rho_m / diag exist on the device).
int32_t fold_build(cosmo_hip_handle* h, const HostCsr& Am, const std::vector<int>& prp, const std::vector<int>& pcol,
                   const std::vector<real>& pval) {
  fold_free(h);
  if (const char* e = getenv("COSMO_HIP_OP_FOLD")) if (e[0] == '0') return COSMO_HIP_OK;
  if (!h->op_split || h->n <= 0 || (!h->cg_sr && !h->cg_ru)) return COSMO_HIP_OK;     // literal CG: needs the {r, u} records; single-reduction CG: its own records
  if (const char* e = getenv("COSMO_HIP_CG_PERSIST")) if (atoi(e)) return COSMO_HIP_OK;     // the single-launch lab path keeps the split operator
  const long long n = h->n;
  const int mm = Am.nrows;
  const long long nnzP = prp.empty() ? 0 : prp[(size_t)n];
  // cost gate: terms = sum over the rows of Am of len^2; the assembled matrix must stay within a small multiple of what the two
  // products of the split operator stream (a 10-nonzero row becomes 100 terms: fine for a few thousand such rows, not for config 2)
  long long nterms = 0;
  for (int r = 0; r < mm; ++r) { const long long len = Am.rowptr[r + 1] - Am.rowptr[r]; nterms += len * len; }
  const long long streamed = 2 * (long long)Am.col.size() + nnzP + n;
  if (nterms > 6 * streamed || nterms > (1LL << 24)) return COSMO_HIP_OK;
  // PARTIAL ASSEMBLY (round 6): a row of `len` nonzeros costs len^2 assembled entries but 2 len entries if it stays factored -- one entry of Ad' rho in the
  // stored matrix, gathered from the record {(Ad r)_kd, (Ad u_prev)_kd}, and one entry of Ad for the linear update of (Ad r) in k_cg_updF.  Rows with
  // >= 4 nonzeros stay factored when that at least halves the stored entries (BASELINE config 5: 6 000 ten-nonzero Zero / Nonnegatives rows are 600 k of
  // the 708 k entries of M; partially assembled 107 k + 60 k).  MEASURED (profiles/r06_partial_assembly.txt): k_cg_dirM 7.7 -> 4.9 us -- it is then at the
  // floor of a kernel of this chain, k_cg_upd<false> with next to no data takes the same 4.9 us -- but k_cg_updF 4.9 -> 6.2 us (the dense rows' own
  // col -> gather chain sits BEHIND alpha); per Krylov iteration as the loop enqueues it 11.72 -> 11.30 us, 250.9 -> 253.9 it/s (+1.2 %): below what it costs in
  // code paths, hence OPT-IN (COSMO_HIP_FOLD_FACTOR=1; COSMO_HIP_FOLD_FACTOR_MIN=len moves the row-length threshold).  The literal recurrence only.
  std::vector<int> did((size_t)mm, -1);
  int nd = 0;
  { bool factor = false;                           // OPT-IN (COSMO_HIP_FOLD_FACTOR=1): measured +1.2 % on BASELINE config 5, see below
    int lmin = 4;
    if (const char* e = getenv("COSMO_HIP_FOLD_FACTOR")) factor = !h->cg_sr && !h->cg_jacobi && atoi(e) != 0;
    if (const char* e = getenv("COSMO_HIP_FOLD_FACTOR_MIN")) { const int v = atoi(e); if (v >= 2) lmin = v; }
    if (factor) {
      long long part = 0, dense_nnz = 0;
      for (int r = 0; r < mm; ++r) { const long long len = Am.rowptr[r + 1] - Am.rowptr[r]; if (len >= lmin) { dense_nnz += len; } else part += len * len; }
      if (dense_nnz > 0 && 2 * (part + 2 * dense_nnz) <= nterms)
        for (int r = 0; r < mm; ++r) if (Am.rowptr[r + 1] - Am.rowptr[r] >= lmin) did[(size_t)r] = nd++;
    } }
  // Am' as lists (row of Am, value) per column
  std::vector<int> tp((size_t)n + 1, 0);
  for (int cidx : Am.col) tp[(size_t)cidx + 1]++;
  for (long long j = 0; j < n; ++j) tp[j + 1] += tp[j];
  std::vector<int> trw(Am.col.size());
  std::vector<real> tvl(Am.col.size());
  { std::vector<int> pos(tp.begin(), tp.end() - 1);
    for (int r = 0; r < mm; ++r) for (int k = Am.rowptr[r]; k < Am.rowptr[r + 1]; ++k) { const int p = pos[Am.col[k]]++; trw[p] = r; tvl[p] = Am.val[k]; } }
  HostCsr M;
  M.nrows = (int)n; M.ncols = (int)n; M.rowptr.assign((size_t)n + 1, 0);
  std::vector<real> base;
  std::vector<int> drow, tptr, trow;
  std::vector<real> tprod;
  tptr.push_back(0);
  struct Term { int j, k; real prod; };
  std::vector<Term> terms;
  std::vector<int> fullcols;
  long long nnz_full = 0;
  for (long long i = 0; i < n; ++i) {
    terms.clear();
    for (int q = tp[i]; q < tp[i + 1]; ++q) {
      const int k = trw[q];
      const real aki = tvl[q];
      if (did[(size_t)k] >= 0) continue;                          // a dense row: stays factored (its entry of Ad' is appended behind the columns < n below)
      for (int p = Am.rowptr[k]; p < Am.rowptr[k + 1]; ++p) terms.push_back({Am.col[p], k, aki * Am.val[p]});
    }
    if (nd > 0) {                                                 // nonzeros row i of the FULLY assembled operator would have (the unit of the bench's algorithmic bytes)
      fullcols.clear();
      fullcols.push_back((int)i);
      if (!prp.empty()) for (int p = prp[i]; p < prp[i + 1]; ++p) fullcols.push_back(pcol[p]);
      for (int q = tp[i]; q < tp[i + 1]; ++q) { const int k = trw[q]; for (int p = Am.rowptr[k]; p < Am.rowptr[k + 1]; ++p) fullcols.push_back(Am.col[p]); }
      std::sort(fullcols.begin(), fullcols.end());
      nnz_full += (long long)(std::unique(fullcols.begin(), fullcols.end()) - fullcols.begin());
    }
    std::sort(terms.begin(), terms.end(), [](const Term& a, const Term& b) { return a.j != b.j ? a.j < b.j : a.k < b.k; });
    // three-way merge over ascending columns: P row i, the term columns, the diagonal
    size_t it = 0;
    int ip = prp.empty() ? 0 : prp[i];
    const int ipe = prp.empty() ? 0 : prp[i + 1];
    bool diag_done = false;
    for (;;) {
      int j = INT32_MAX;
      if (it < terms.size()) j = std::min(j, terms[it].j);
      if (ip < ipe) j = std::min(j, pcol[ip]);
      if (!diag_done) j = std::min(j, (int)i);
      if (j == INT32_MAX) break;
      real b = 0.0;
      while (ip < ipe && pcol[ip] == j) { b += pval[ip]; ++ip; }      // duplicated entries of P (the C ABI passes them through) are summed, as the SpMV kernels do
      while (it < terms.size() && terms[it].j == j) { trow.push_back(terms[it].k); tprod.push_back(terms[it].prod); ++it; }
      if (j == (int)i) diag_done = true;
      M.col.push_back(j); M.val.push_back(0.0);
      base.push_back(b); drow.push_back(j == (int)i ? (int)i : -1);
      tptr.push_back((int)trow.size());
    }
    // the factored part of row i: one entry a_ki rho_k per dense row k that holds column i, at column n + did[k] (ascending: the lists are in Am-row order)
    for (int q = tp[i]; q < tp[i + 1]; ++q) {
      const int k = trw[q];
      if (did[(size_t)k] < 0) continue;
      M.col.push_back((int)n + did[(size_t)k]); M.val.push_back(0.0);
      base.push_back(R(0.0)); drow.push_back(-1);
      trow.push_back(k); tprod.push_back(tvl[q]);
      tptr.push_back((int)trow.size());
    }
    M.rowptr[i + 1] = (int)M.col.size();
  }
  M.ncols = (int)n + nd;
  if ((long long)M.col.size() >= 2147483647LL) return COSMO_HIP_OK;
  FoldPlan* f = new FoldPlan();
  h->fold = f;
  f->nterms = (long long)trow.size();
  f->nd = nd;
  f->nnz_full = nd > 0 ? nnz_full : (long long)M.col.size();
  if (nd > 0) {                                                   // the dense rows themselves (values without rho) + their records
    HostCsr Adh;
    Adh.nrows = nd; Adh.ncols = (int)n; Adh.rowptr.assign((size_t)nd + 1, 0);
    for (int r = 0; r < mm; ++r) {
      if (did[(size_t)r] < 0) continue;
      for (int p = Am.rowptr[r]; p < Am.rowptr[r + 1]; ++p) { Adh.col.push_back(Am.col[p]); Adh.val.push_back(Am.val[p]); }
      Adh.rowptr[(size_t)did[(size_t)r] + 1] = (int)Adh.col.size();
    }
    CHK(upload_csr(h, Adh, f->Ad, (int)n, ADSL * COSMO_BS));
    // the {r, u} records by iteration parity (k_cg_updF: owners write the new ones while the dense rows gather the old ones)
    if (h->cg_ru) { (void)hipFree(h->cg_ru); h->cg_ru = nullptr; }
    HIPCHK(h, hipMalloc((void**)&h->cg_ru, sizeof(real) * 4 * (size_t)n));
    HIPCHK(h, hipMemset(h->cg_ru, 0, sizeof(real) * 4 * (size_t)n));
    HIPCHK(h, hipMalloc((void**)&f->tt, sizeof(real2) * (size_t)nd));
    HIPCHK(h, hipMalloc((void**)&f->tcur, sizeof(real) * (size_t)nd));
    HIPCHK(h, hipMalloc((void**)&f->tx, sizeof(real) * (size_t)nd));
    HIPCHK(h, hipMemset(f->tt, 0, sizeof(real2) * (size_t)nd)); HIPCHK(h, hipMemset(f->tcur, 0, sizeof(real) * (size_t)nd)); HIPCHK(h, hipMemset(f->tx, 0, sizeof(real) * (size_t)nd));
  }
  // Tile size: measured on BASELINE config 5 (708 k nonzeros; profiles/r02_cfg5_fold_tile_sweep.txt) 256 / 384 / 512 / 768 / 1408
  // nonzeros per tile all land within +-2 % (144-150 it/s): the kernel is a chain of dependent round trips, not a stream.  768 keeps
  // the register-staged slots at four per thread and the tile count under the partial-slot limit for operators up to ~1.5 M nonzeros.
  int tile = 768;
  if ((long long)M.col.size() > 700LL * COSMO_MAX_PARTIALS) tile = 0;      // large operators: the size heuristic of build_row_blocks
  if (const char* e = getenv("COSMO_HIP_FOLD_TILE")) { const int v = atoi(e); if (v >= 64 && v <= COSMO_NNZ_PER_BLOCK) tile = v; }
  CHK(upload_csr(h, M, f->M, (int)n, tile));
  f->slots = tile == 0 ? 8 : tile <= COSMO_BS ? 1 : tile <= 2 * COSMO_BS ? 2 : tile <= 3 * COSMO_BS ? 3 : tile <= 4 * COSMO_BS ? 4 : 8;
  // tile-major padded copy of (col, val) -- fixed-size tiles only.  OPT-IN (COSMO_HIP_FOLD_PAD=1): measured on BASELINE config 5, two alternating
  // pairs of runs: 12.35 / 12.35 us per Krylov iteration with the copy against 12.27 / 12.30 us without (237.3 / 236.9 vs 237.7 / 237.9 it/s) -- the
  // descriptor's round trip was not on the critical path (the gather behind (col, val) and the partial-sum reduction are); bit-identical either way
  { bool pad = false;
    if (const char* e = getenv("COSMO_HIP_FOLD_PAD")) pad = tile != 0 && f->slots <= 4 && atoi(e) != 0;
    if (pad) {
      const int cap = f->slots * COSMO_BS;
      std::vector<int> rbnd, rbd;
      build_row_blocks(M.rowptr, M.nrows, rbnd, tile);                       // the same boundaries upload_csr used
      const size_t nbk = rbnd.size() - 1;
      if ((long long)nbk == f->M.nb && nbk * (size_t)cap < (size_t)1 << 30) {
        std::vector<int> pc(nbk * cap, 0), pp(M.col.size(), 0);
        std::vector<real> pv(nbk * cap, R(0.0));
        for (size_t k = 0; k < nbk; ++k) {
          const int z0 = M.rowptr[rbnd[k]], z1 = M.rowptr[rbnd[k + 1]];
          for (int e2 = z0; e2 < z1; ++e2) {
            const size_t q = (z1 - z0 <= cap) ? k * cap + (size_t)(e2 - z0) : k * cap;     // a single long row (generic path) is not streamed from the copy
            if (z1 - z0 <= cap) { pc[q] = M.col[e2]; pv[q] = M.val[e2]; }
            pp[(size_t)e2] = (int)q;
          }
        }
        CHK(up(h, &f->pcol, pc)); CHK(up(h, &f->pval, pv)); CHK(up(h, &f->ppos, pp));
        f->cap = cap;
      }
    } }
  { std::vector<int> dpos((size_t)n, 0);                        // every row has its diagonal entry (the three-way merge adds it)
    for (size_t pz = 0; pz < drow.size(); ++pz) if (drow[pz] >= 0) dpos[(size_t)drow[pz]] = (int)pz;
    CHK(up(h, &f->dpos, dpos));
    HIPCHK(h, hipMalloc((void**)&f->dinv, sizeof(real) * (size_t)n)); }
  CHK(up(h, &f->base, base)); CHK(up(h, &f->drow, drow)); CHK(up(h, &f->tptr, tptr)); CHK(up(h, &f->trow, trow)); CHK(up(h, &f->tprod, tprod));
  h->op_fold = true;
  return COSMO_HIP_OK;
}

int32_t fold_refresh(cosmo_hip_handle* h) {
  FoldPlan* f = (FoldPlan*)h->fold;
  if (!h->op_fold || !f) return COSMO_HIP_OK;
  hipLaunchKernelGGL(k_fold_refresh, dim3(ew_grid(f->M.nnz)), dim3(COSMO_BS), 0, h->stream, f->M.nnz, f->base, f->drow, f->tptr, f->trow,
                     f->tprod, h->op_rho_m, h->op_diag, h->prm.sigma, f->M.val, (const int*)f->ppos, f->pval);
  if (h->cg_jacobi)
    hipLaunchKernelGGL(k_fold_dinv, dim3(ew_grid(h->n)), dim3(COSMO_BS), 0, h->stream, h->n, f->dpos, f->M.val, f->dinv);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t fold_enqueue_start(cosmo_hip_handle* h, int guard, real tol_k) {
  FoldPlan* f = (FoldPlan*)h->fold;
  const int gD = f->nd > 0 ? std::max(f->Ad.grid, 1) : 0;
  prof_begin(h, KC_OP_APPLY);
  if (f->nd > 0)               // partial assembly: (Ad x) for the start residual r0 = rhs - (Ms x + Ad' rho (Ad x))
    hipLaunchKernelGGL(k_ad_dot, dim3(gD), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(f->Ad), (const real*)h->x_tl, f->tx, (real2*)nullptr);
  if (h->cg_jacobi)
    hipLaunchKernelGGL(k_fold_start<true>, dim3(f->M.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(f->M), h->x_tl, h->rhs, h->r,
                       (real2*)h->cg_ru, PARTS(h, SLOT_RR), PARTS(h, SLOT_BB), h->n_bb, tol_k, (const real*)f->dinv, PARTS(h, SLOT_AUX2), (const real*)nullptr);
  else
    hipLaunchKernelGGL(k_fold_start<false>, dim3(f->M.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(f->M), h->x_tl, h->rhs, h->r,
                       (real2*)h->cg_ru, PARTS(h, SLOT_RR), PARTS(h, SLOT_BB), h->n_bb, tol_k, (const real*)nullptr, (real*)nullptr, (const real*)(f->nd > 0 ? f->tx : nullptr));
  if (f->nd > 0)               // ... and the records {Ad r0, 0} of iteration 0
    hipLaunchKernelGGL(k_ad_dot, dim3(gD), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(f->Ad), (const real*)h->r, (real*)nullptr, f->tt);
  prof_end(h);
  h->spmv_calls[0] += 1; h->spmv_calls[1] += 2; h->spmv_calls[2] += 1;     // the reference's multiplication count (A' y2 of the rhs + reduced_mul! = A, A', P)
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// par: parity of the iteration (factored operators keep the {r, u} records by parity; a launch of the captured chain gets it as an argument because its k is
// read on the device AFTER the gathers have been requested)
static void fold_launch_pair(cosmo_hip_handle* h, FoldPlan* f, int guard, int k, int n_rr, int check_first = 0, int par = 0) {
  const long long n = h->n;
  if (k >= 0) par = k & 1;
  const real2* ru_cur = (const real2*)h->cg_ru + (f->nd > 0 ? (size_t)par * n : 0);
  real2* ru_nxt = (real2*)h->cg_ru + (f->nd > 0 ? (size_t)(par ^ 1) * n : 0);
  prof_begin(h, KC_OP_APPLY);
#define LAUNCH_DIRM_(SLN, PCF) hipLaunchKernelGGL((k_cg_dirM<SLN, PCF>), dim3(f->M.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, check_first, k, n, n, \
                         PARTS(h, SLOT_RR), n_rr, view_of(f->M), ru_cur, h->c, h->u, PARTS(h, SLOT_UC), (const real*)PARTS(h, SLOT_AUX2), \
                         (const int*)(f->cap == SLN * COSMO_BS ? f->pcol : nullptr), (const real*)f->pval, (const real2*)f->tt, f->tcur, f->nd)
#define LAUNCH_DIRM(SLN) do { if (h->cg_jacobi) LAUNCH_DIRM_(SLN, true); else LAUNCH_DIRM_(SLN, false); } while (0)
  switch (f->slots) {
    case 1: LAUNCH_DIRM(1); break;
    case 2: LAUNCH_DIRM(2); break;
    case 3: LAUNCH_DIRM(3); break;
    case 4: LAUNCH_DIRM(4); break;
    default: LAUNCH_DIRM(8); break;
  }
#undef LAUNCH_DIRM
#undef LAUNCH_DIRM_
  prof_end(h);
  if (f->nd > 0) {             // partial assembly: the vector update + the linear update of (Ad r) on the dense rows (workgroups behind the first gE)
    const int gE = ew_grid(n), gD = std::max(f->Ad.grid, 1);
    prof_begin(h, KC_CG_UPD);
    hipLaunchKernelGGL(k_cg_updF, dim3(gE + gD), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, n, PARTS(h, SLOT_UC), f->M.grid, h->u, h->c, h->x_tl, h->r, PARTS(h, SLOT_RR),
                       ru_cur, ru_nxt, gE, f->nd, view_of(f->Ad), f->tt, (const real*)f->tcur);
    prof_end(h);
  } else (void)launch_cg_upd(h, guard, k, f->M.grid);
}

// The speculative Krylov iterations of a solve in the loop (k = 1 .. budget - 1, all with the same arguments once the iteration index is
// read on the device) are launched as a CAPTURED CHAIN of `chain_len` iterations: a direct launch costs the host 3.4-3.8 us, a kernel of a
// captured chain 0.03-0.3 us (profiles/r03_launch_cost_direct_vs_graph.txt), and BASELINE config 5 issues ~ 400 such launches per ADMM
// iteration -- the kernel timeline showed the launching thread falling behind inside the Krylov loop (12 % idle) and a loaded host cost
// 13 % of the throughput.  Index protocol: k_cg_dirM reads ctl->cg_k (written by the k_cg_upd in front of it) and publishes it as
// ctl->cg_kd for the k_cg_upd behind it, which writes cg_k = k + 1; the final check reads cg_k.  Arithmetic and launch order are those of
// the direct path (COSMO_HIP_CG_GRAPH=0; tests/test_gpu_cg_fold.py compares the two bit for bit), and so is the number of iterations enqueued:
// whole chains first, the remainder of the budget directly with the index as a kernel argument.
static bool fold_chain_ready(cosmo_hip_handle* h, FoldPlan* f) {
  if (f->chain_off || h->profiling) return false;
  if (f->chain) return true;
  if (const char* e = getenv("COSMO_HIP_CG_GRAPH")) { if (atoi(e) == 0) { f->chain_off = 1; return false; } }
  int len = 16;
  if (const char* e = getenv("COSMO_HIP_CG_GRAPH_LEN")) { const int v = atoi(e); if (v >= 1 && v <= 256) len = v; }
  if (f->nd > 0) len += len & 1;                 // records by parity: the chain starts at an odd iteration and must end on an even one
  const int gE = ew_grid(h->n);
  hipGraphExec_t ex[2] = {nullptr, nullptr};
  for (int cf = 0; cf < 2; ++cf) {               // the chain, and the chain of iterations that are expected to be no-ops
    hipGraph_t g = nullptr;
    bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      for (int i = 0; i < len; ++i) fold_launch_pair(h, f, 1, -1, gE, cf, (i + 1) & 1);
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
  f->chain = ex[0]; f->chain_cf = ex[1]; f->chain_len = len;
  return true;
}

int32_t fold_enqueue_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count) {
  FoldPlan* f = (FoldPlan*)h->fold;
  const int gE = ew_grid(h->n);
  int k = k_begin;
  const int k_end = k_begin + count;
  if (guard == 1 && k_begin == 0 && count > 1 && fold_chain_ready(h, f)) {
    fold_launch_pair(h, f, guard, 0, f->M.grid);                 // k = 0 reads the partials of k_fold_start: its own launch
    h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
    for (k = 1; k + f->chain_len <= k_end; k += f->chain_len) {  // whole chains; the remainder (< chain_len iterations) goes out directly below
      HIPCHK(h, hipGraphLaunch((hipGraphExec_t)(k >= h->cg_k_likely ? f->chain_cf : f->chain), h->stream));
      h->spmv_calls[0] += f->chain_len; h->spmv_calls[1] += f->chain_len; h->spmv_calls[2] += f->chain_len;
    }
  }
  for (; k < k_end; ++k) {
    fold_launch_pair(h, f, guard, k, (k == 0) ? f->M.grid : gE, (k >= h->cg_k_likely) ? 1 : 0);
    h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  }
  CHK(launch_cg_dir_check(h, guard, k_end, (k_end == 0) ? f->M.grid : gE));
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// diagnostics: out = {enabled, nnz of the FULLY assembled M, terms, tiles, rows of Am kept factored (partial assembly; 0 = none), stored entries of [Ms | Ad' rho]}
extern "C" int32_t cosmo_hip_fold_stats(cosmo_hip_handle* h, int64_t out[6]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 6; ++i) out[i] = 0;
  FoldPlan* f = (FoldPlan*)h->fold;
  if (!h->op_fold || !f) return COSMO_HIP_OK;
  out[0] = 1; out[1] = f->nnz_full; out[2] = f->nterms; out[3] = f->M.nb; out[4] = f->nd; out[5] = f->M.nnz;
  return COSMO_HIP_OK;
}
