// minres.hip -- device-resident MINRES for the two MINRES KKT plugins of the reference:
//   COSMO_HIP_KKT_MINRES_REDUCED : IndirectReducedKKTSolver(solver_type = :MINRES)  (kktsolver_indirect.jl:3-88)
//                                  (P + sigma I + A' rho A) y1 = x1 + A' rho x2 ,  y2 = rho (A y1 - x2)
//   COSMO_HIP_KKT_MINRES         : MINRESIndirectKKTSolver = IndirectKKTSolver       (kktsolver_indirect.jl:90-162)
//                                  [P + sigma I, A'; A, -1/rho] y = x   (n+m unknowns)
// Both call IterativeSolvers.jl v0.9 `minres!(x, L, b; abstol = tol_k / ||L x0 - b||, reltol = 0)` with a warm start
// (kktsolver_indirect.jl:72-73, 151-152).  The algorithm below restates minres.jl of that package (Lanczos three-term
// recurrence + Givens rotations) -- the package is not vendored in the reference tree, so parity for
// this arithmetic is pinned on a dense solve, as the reference's own (disabled) test does (test/UnitTests/kktsolver.jl:97-109).
//
// Per iteration: operator apply (two CSR-stream SpMV kernels) fused with the v_prev subtraction and the <v_curr, v_next>
// partials; one orthogonalisation kernel (||v_next||^2 partials); one update kernel that folds the scalar recurrences
// (every workgroup recomputes them from a parity-double-buffered state, workgroup 0 publishes the next state).
#include "device_utils.h"
#include <algorithm>
#include <math.h>

int32_t enqueue_tail(cosmo_hip_handle* h, int loop_mode);
int32_t enqueue_count_solve(cosmo_hip_handle* h);
int32_t sync_ctl(cosmo_hip_handle* h);
int32_t launch_spmv_A_rho(cosmo_hip_handle* h, int guard, int mode, const real* v, real* out);
int32_t launch_reduced_rhs(cosmo_hip_handle* h, int guard, real* out_rhs);
int32_t enqueue_y2_only(cosmo_hip_handle* h);

// state slot layout inside Ctl::minres (two slots of 8 doubles, indexed by iteration parity)
enum { MS_H1 = 0, MS_CP, MS_SP, MS_CC, MS_SC, MS_RHS0, MS_RES, MS_PAD };

struct MrVecs {
  real *v[3], *w[3];   // Krylov basis and W = V R^-1 recurrences (rotating)
  real* x;             // solution / warm start (n or n+m)
  real* b;             // right-hand side of the (reduced or full) system
  long long N;
};

// Givens rotation as LinearAlgebra.givensAlgorithm(f, g) for reals: [c s; -s c] [f; g] = [r; 0]
__device__ __forceinline__ void givens(real f, real g, real& c, real& s, real& r) {
  if (g == R(0.0)) { c = R(1.0); s = R(0.0); r = f; return; }
  if (f == R(0.0)) { c = R(0.0); s = R(1.0); r = g; return; }
  r = hypot(f, g);
  c = f / r; s = g / r;
  if (fabs(f) > fabs(g) && c < R(0.0)) { c = -c; s = -s; r = -r; }
}

// top block rows (0..n): y = P v1 + (sigma v1 + A' v2) through the merged operator [P | A'].
// mode 0 (start): vout = b - y (top part), partial ||.||^2        mode 1 (iteration): y -= h1 vprev ; vout = y ; partial <vcurr, y>
__global__ __launch_bounds__(COSMO_BS) void k_mr_op_top(Ctl* __restrict__ ctl, int guard, int mode, int it, long long maxiter, CsrView PT, real sigma,
                                                        const real* __restrict__ v1, const real* __restrict__ v2,
                                                        const real* __restrict__ vprev, const real* __restrict__ b,
                                                        real* __restrict__ vout, real* __restrict__ part, const real* __restrict__ diag) {
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  const real h1 = ctl->minres[((it - 1) & 1) * 8 + MS_H1];
  if (mode == 1) {
    // done(m, iteration): iteration > maxiter || resnorm <= tolerance, evaluated BEFORE the iteration.  Every workgroup
    // takes the same decision from the read-only previous state; workgroup 0 publishes it for the later kernels.
    const real res = ctl->minres[((it - 1) & 1) * 8 + MS_RES];
    if (res <= ctl->tol || (long long)it > maxiter) {
      if (blockIdx.x == 0 && threadIdx.x == 0) ctl->cg_done = 1;
      return;
    }
  }
  real acc = 0.0;
  for (int k = blockIdx.x; k < PT.nb; k += gridDim.x) {
    csr_stream_tile(PT, v1, v2, k, lds, red, [&](int row, real s1, real s2) {
      const real vc = v1[row];
      real y = s1 + (sigma * vc + s2);
      if (diag) y += diag[row] * vc;           // split operator (row-sharded handles): the singleton rows of A are the diagonal of A' rho A
      if (mode == 0) {
        const real r = b[row] - y;
        vout[row] = r;
        acc += r * r;
      } else {
        if (it > 1) y = y - h1 * vprev[row];
        vout[row] = y;
        acc += vc * y;
      }
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// bottom block rows of the full KKT operator: y2 = A v1 + (-v2 / rho)          (kktsolver_indirect.jl:142-144)
__global__ __launch_bounds__(COSMO_BS) void k_mr_op_bot(Ctl* __restrict__ ctl, int guard, int mode, int it, CsrView A, long long n,
                                                        const real* __restrict__ v1, const real* __restrict__ v2,
                                                        const real* __restrict__ rho, const real* __restrict__ vprev,
                                                        const real* __restrict__ b, real* __restrict__ vout,
                                                        real* __restrict__ part) {
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  const real h1 = ctl->minres[((it - 1) & 1) * 8 + MS_H1];
  real acc = 0.0;
  for (int k = blockIdx.x; k < A.nb; k += gridDim.x) {
    csr_stream_tile(A, v1, v1, k, lds, red, [&](int row, real s1, real s2) {
      const real vc = v2[row];
      real y = (s1 + s2) + (-vc / rho[row]);
      if (mode == 0) {
        const real r = b[n + row] - y;
        vout[n + row] = r;
        acc += r * r;
      } else {
        if (it > 1) y = y - h1 * vprev[n + row];
        vout[n + row] = y;
        acc += vc * y;
      }
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// start: resnorm = ||b - L x0|| ; tolerance = tol_k / resnorm (abstol, reltol = 0) ; v_curr /= resnorm ; state init
__global__ __launch_bounds__(COSMO_BS) void k_mr_start(Ctl* __restrict__ ctl, int guard, long long N, const real* __restrict__ part,
                                                       int npart, real tol_k, real* __restrict__ vcurr) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real rr = reduce_partials_sum(part, npart, red);
  const real res = sqrt(rr);
  const real tol = tol_k / res;
  const bool done = (res <= tol) || !(res > R(0.0));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    real* st = ctl->minres;   // slot 0 = state "after iteration 0"
    st[MS_H1] = 0.0; st[MS_CP] = 1.0; st[MS_SP] = 0.0; st[MS_CC] = 1.0; st[MS_SC] = 0.0; st[MS_RHS0] = res; st[MS_RES] = res;
    ctl->tol = tol; ctl->rhs_norm = res;
    ctl->cg_done = done ? 1 : 0;
    ctl->cg_k = 0;
  }
  if (done) return;
  const real inv = R(1.0) / res;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) vcurr[i] = vcurr[i] * inv;
}

// v_next -= <v_curr, v_next> v_curr ; partial ||v_next||^2                     (minres.jl: orthogonalise w.r.t. v_curr)
__global__ __launch_bounds__(COSMO_BS) void k_mr_orth(Ctl* __restrict__ ctl, int guard, long long N, const real* __restrict__ part_in,
                                                      int npart, const real* __restrict__ vcurr, real* __restrict__ vnext,
                                                      real* __restrict__ part_out) {
  const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
  real vc0 = 0.0, vn0 = 0.0;
  if (i0 < N) { vc0 = vcurr[i0]; vn0 = vnext[i0]; }
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real red[COSMO_BS / 64];
  const real proj = reduce_partials_sum(part_in, npart, red);
  real acc = 0.0;
  if (i0 < N) { const real y = vn0 - proj * vc0; vnext[i0] = y; acc += y * y; }
  for (long long i = i0 + (long long)gridDim.x * COSMO_BS; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real y = vnext[i] - proj * vcurr[i];
    vnext[i] = y;
    acc += y * y;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    part_out[blockIdx.x] = acc;
    if (blockIdx.x == 0) ctl->udotc_slot = proj;   // H[3] of this iteration (Julia's 1-based H[3])
  }
}

// scalar recurrences + v_next normalisation + W recurrence + solution update + convergence test for the NEXT iteration
__global__ __launch_bounds__(COSMO_BS) void k_mr_update(Ctl* __restrict__ ctl, int guard, int it, long long N, long long maxiter,
                                                        const real* __restrict__ part_nn, int npart, const real* __restrict__ vcurr,
                                                        real* __restrict__ vnext, const real* __restrict__ wprev,
                                                        const real* __restrict__ wcurr, real* __restrict__ wnext,
                                                        real* __restrict__ x) {
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real red[COSMO_BS / 64];
  const real* so = ctl->minres + ((it - 1) & 1) * 8;
  const real nn = reduce_partials_sum(part_nn, npart, red);
  real H0 = 0.0, H1 = so[MS_H1], H2 = ctl->udotc_slot, H3 = sqrt(nn);
  const real c_prev = so[MS_CP], s_prev = so[MS_SP], c_curr = so[MS_CC], s_curr = so[MS_SC];
  real rhs0 = so[MS_RHS0];
  if (it > 2) { H0 = s_prev * H1; H1 = c_prev * H1; }
  if (it > 1) {
    const real tmp = -s_curr * H1 + c_curr * H2;
    H1 = c_curr * H1 + s_curr * H2;
    H2 = tmp;
  }
  real c, s, r;
  givens(H2, H3, c, s, r);
  H2 = r;
  const real rhs1 = -s * rhs0;
  rhs0 = c * rhs0;
  const real inv_h3 = R(1.0) / H3, inv_h2 = R(1.0) / H2;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real vc = vcurr[i];
    vnext[i] = vnext[i] * inv_h3;
    real wn = vc;
    if (it > 1) wn = wn - H1 * wcurr[i];
    if (it > 2) wn = wn - H0 * wprev[i];
    wn = wn * inv_h2;
    wnext[i] = wn;
    x[i] = x[i] + rhs0 * wn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    real* sn = ctl->minres + (it & 1) * 8;
    sn[MS_H1] = H3; sn[MS_CP] = c_curr; sn[MS_SP] = s_curr; sn[MS_CC] = c; sn[MS_SC] = s; sn[MS_RHS0] = rhs1;
    const real res = fabs(rhs1);
    sn[MS_RES] = res;
    ctl->cg_k = it;   // the stop rule for iteration it+1 is evaluated by the next k_mr_op_top / k_mr_finish (never here:
                      // other workgroups of THIS kernel still read cg_done)
  }
}

// stop rule after the last budgeted iteration (so that the tail / the host see a final cg_done)
__global__ void k_mr_finish(Ctl* ctl, long long maxiter) {
  if (ctl->cg_done) return;
  const int it = ctl->cg_k;
  const real res = ctl->minres[(it & 1) * 8 + MS_RES];
  if (res <= ctl->tol || (long long)(it + 1) > maxiter) ctl->cg_done = 1;
}

// full-KKT tail: sol = [x_tl; nu] is the MINRES iterate itself; s_tl = (2 s - w_s) - nu ./ rho ; w updates (solver.jl:55,63-64)
__global__ __launch_bounds__(COSMO_BS) void k_mr_tail_full(Ctl* __restrict__ ctl, int loop_mode, long long n, long long m, real alpha,
                                                           const real* __restrict__ xsol, const real* __restrict__ rho,
                                                           const real* __restrict__ s, real* __restrict__ x_tl,
                                                           real* __restrict__ nu, real* __restrict__ s_tl, real* __restrict__ w) {
  if (loop_mode) {
    if (ctl->halt) return;
    if (!ctl->cg_done) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->stalled = 1; ctl->halt = 1; }
      return;
    }
  }
  const long long N = n + m;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real v = xsol[i];
    if (i < n) {
      x_tl[i] = v;
      if (loop_mode) { const real wv = w[i]; w[i] = wv + alpha * (v - wv); }
    } else {
      const long long r = i - n;
      nu[r] = v;
      if (loop_mode) {
        const real sv = s[r], wv = w[i];
        const real st = (R(2.0) * sv - wv) - v / rho[r];
        s_tl[r] = st;
        w[i] = wv + alpha * (st - sv);
      }
    }
  }
  if (loop_mode && blockIdx.x == 0 && threadIdx.x == 0) {
    const int k = ctl->cg_k;
    ctl->iter += 1; ctl->solves += 1; ctl->kkt_iters_total += k;
    if (k > ctl->cg_k_max) ctl->cg_k_max = k;
  }
}

// reduced system right-hand side: b = A' (rho .* ls_s) + ls_x is produced by k_cg_rhs (kernels.hip); copy helpers below
__global__ __launch_bounds__(COSMO_BS) void k_mr_copy2(long long n, long long m, const real* __restrict__ a, const real* __restrict__ b2,
                                                       real* __restrict__ out) {
  const long long N = n + m;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS)
    out[i] = (i < n) ? a[i] : b2[i - n];
}

// =====================================================================================================================
static inline int ewg(long long N) {
  long long g = (N + COSMO_BS - 1) / COSMO_BS;
  if (g < 1) g = 1;
  if (g > COSMO_MAX_PARTIALS) g = COSMO_MAX_PARTIALS;
  return (int)g;
}
#define MPARTS(h, slot) ((h)->partials + (size_t)(slot) * COSMO_MAX_PARTIALS)

int32_t minres_alloc(cosmo_hip_handle* h) {
  if (h->mr) { (void)hipFree(h->mr); h->mr = nullptr; }
  const size_t N = (size_t)(h->n + h->m);
  HIPCHK(h, hipMalloc((void**)&h->mr, sizeof(real) * 8 * std::max<size_t>(N, 1)));
  HIPCHK(h, hipMemsetAsync(h->mr, 0, sizeof(real) * 8 * std::max<size_t>(N, 1), h->stream));
  return COSMO_HIP_OK;
}

static MrVecs vecs_of(cosmo_hip_handle* h) {
  MrVecs V;
  const bool full = h->prm.kkt_kind == COSMO_HIP_KKT_MINRES;
  const long long NN = h->n + h->m;
  V.N = full ? NN : h->n;
  for (int i = 0; i < 3; ++i) { V.v[i] = h->mr + (size_t)i * NN; V.w[i] = h->mr + (size_t)(3 + i) * NN; }
  V.x = h->mr + (size_t)6 * NN;       // previous_solution (persists across solves: warm start)
  V.b = h->mr + (size_t)7 * NN;
  return V;
}

// operator apply y = L v for either system; partial slots: SLOT_UC (top) and SLOT_AUX1 (bottom, full system only)
static int32_t enqueue_mr_apply(cosmo_hip_handle* h, int guard, int mode, int it, const MrVecs& V, const real* v, const real* vprev,
                                real* vout) {
  const bool full = h->prm.kkt_kind == COSMO_HIP_KKT_MINRES;
  const long long n = h->n;
  prof_begin(h, KC_OP_APPLY);
  if (!full) {
    // tmp_m = rho .* (A v) ; out = [P | A'] [v; tmp_m] + sigma v
    CHK(launch_spmv_A_rho(h, guard, 1, v, h->tmp_m));
    const CsrDev& PTo = h->row_shard ? h->PTm : h->PT;
    hipLaunchKernelGGL(k_mr_op_top, dim3(PTo.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, mode, it, V.N, view_of(PTo), h->prm.sigma, v,
                       h->tmp_m, vprev, V.b, vout, MPARTS(h, SLOT_UC), (const real*)(h->row_shard ? h->op_diag : nullptr));
  } else {
    hipLaunchKernelGGL(k_mr_op_top, dim3(h->PT.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, mode, it, V.N, view_of(h->PT), h->prm.sigma, v,
                       v + n, vprev, V.b, vout, MPARTS(h, SLOT_UC), (const real*)nullptr);
    hipLaunchKernelGGL(k_mr_op_bot, dim3(h->A.grid > 0 ? h->A.grid : 1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, mode, it, view_of(h->A), n,
                       v, v + n, h->rho, vprev, V.b, vout, MPARTS(h, SLOT_UC) + h->PT.grid);
  }
  prof_end(h);
  h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

static int npart_apply(const cosmo_hip_handle* h) {
  if (h->row_shard) return h->PTm.grid;                  // reduced MINRES on the split operator
  return h->PT.grid + ((h->prm.kkt_kind == COSMO_HIP_KKT_MINRES) ? (h->A.grid > 0 ? h->A.grid : 1) : 0);
}

int32_t minres_enqueue_iterations(cosmo_hip_handle* h, int guard, int it_begin, int count) {
  const MrVecs V = vecs_of(h);
  const int gE = ewg(V.N);
  for (int it = it_begin; it < it_begin + count; ++it) {
    // buffers rotate: curr = (it-1) % 3, next = it % 3, prev = (it-2) % 3  (it is 1-based)
    real* vc = V.v[(it + 2) % 3]; real* vn = V.v[it % 3]; real* vp = V.v[(it + 1) % 3];
    real* wc = V.w[(it + 2) % 3]; real* wn = V.w[it % 3]; real* wp = V.w[(it + 1) % 3];
    CHK(enqueue_mr_apply(h, guard, 1, it, V, vc, vp, vn));
    prof_begin(h, KC_MINRES_VEC);
    hipLaunchKernelGGL(k_mr_orth, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, V.N, MPARTS(h, SLOT_UC), npart_apply(h), vc, vn,
                       MPARTS(h, SLOT_RR));
    hipLaunchKernelGGL(k_mr_update, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, it, V.N, V.N, MPARTS(h, SLOT_RR), gE, vc, vn, wp, wc,
                       wn, V.x);
    prof_end(h);
  }
  hipLaunchKernelGGL(k_mr_finish, dim3(1), dim3(1), 0, h->stream, h->ctl, V.N);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

static int32_t enqueue_mr_tail(cosmo_hip_handle* h, int loop_mode) {
  const MrVecs V = vecs_of(h);
  if (h->prm.kkt_kind == COSMO_HIP_KKT_MINRES) {
    prof_begin(h, KC_TAIL);
    hipLaunchKernelGGL(k_mr_tail_full, dim3(ewg(h->n + h->m)), dim3(COSMO_BS), 0, h->stream, h->ctl, loop_mode, h->n, h->m, h->prm.alpha,
                       V.x, h->rho, h->s, h->x_tl, h->nu, h->s_tl, h->w);
    prof_end(h);
    HIPCHK(h, hipGetLastError());
    return COSMO_HIP_OK;
  }
  // reduced: y1 = solution ; y2 = rho (A y1 - x2) and the rest of admm_x!/admm_w! are the CG tail kernel
  HIPCHK(h, hipMemcpyAsync(h->x_tl, V.x, sizeof(real) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
  return enqueue_tail(h, loop_mode);
}

int32_t minres_resume(cosmo_hip_handle* h, int extra) {
  CHK(minres_enqueue_iterations(h, 1, h->ctl_host->cg_k + 1, extra));
  return enqueue_mr_tail(h, 1);
}

// One solve! of either MINRES plugin.  from_loop: ls_x / ls_s / y2 were produced by k_rhs; else they were uploaded and
// y2 = rho .* ls_s has been formed by the caller.
int32_t minres_enqueue_solve(cosmo_hip_handle* h, int guard, bool from_loop) {
  const MrVecs V = vecs_of(h);
  const bool full = h->prm.kkt_kind == COSMO_HIP_KKT_MINRES;
  const real tol_k = h->prm.tol_constant / pow((real)(h->host_solves + 1), h->prm.tol_exponent);
  if (!from_loop) CHK(enqueue_y2_only(h));
  if (full) {
    hipLaunchKernelGGL(k_mr_copy2, dim3(ewg(h->n + h->m)), dim3(COSMO_BS), 0, h->stream, h->n, h->m, h->ls_x, h->ls_s, V.b);
  } else {
    CHK(launch_reduced_rhs(h, guard, V.b));
  }
  // v_curr = b - L x0 (into v[0]) ; the k_rhs / k_y2_only kernels have reset cg_done / cg_k
  CHK(enqueue_mr_apply(h, guard, 0, 1, V, V.x, V.x, V.v[0]));
  hipLaunchKernelGGL(k_mr_start, dim3(ewg(V.N)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, V.N, MPARTS(h, SLOT_UC), npart_apply(h), tol_k,
                     V.v[0]);
  HIPCHK(h, hipGetLastError());
  if (!guard || h->exact_launches) {
    // synchronous pacing (fine-grained entry point / measurement mode)
    int it = 1, chunk = std::max(h->budget, 4);
    for (;;) {
      CHK(sync_ctl(h));
      if (h->ctl_host->cg_done || (guard && h->ctl_host->halt)) break;
      CHK(minres_enqueue_iterations(h, guard, it, chunk));
      it += chunk;
    }
  } else {
    CHK(minres_enqueue_iterations(h, guard, 1, h->budget));
  }
  CHK(enqueue_mr_tail(h, from_loop ? 1 : 0));
  if (!from_loop) CHK(enqueue_count_solve(h));
  return COSMO_HIP_OK;
}
