// infeas.hip -- primal / dual infeasibility certificates (src/infeasibility.jl:1-68, scheduled by src/solver.jl:145-148,
// 326-349).  Every 40 iterations (check_infeasibility) the reference captures delta_y at the top of the next iteration,
// and after that iteration tests
//   primal:  ||E dy||_inf > eps ; ||Dinv A'dy||_inf <= eps ||E dy||_inf ; support function of K at -dy/||.|| minus <dyn, b> <= eps
//   dual  :  ||D dx||_inf > eps ; <q,dx>/(||D dx|| c) < -eps ; ||Dinv P dx||_inf/(||D dx|| c) <= eps ; Einv A dx / ||D dx|| in K_polar_recc
// All O(n+m+nnz) ingredients (the three SpMVs, the scaled norms, the dots, the per-cone membership tests incl. the PSD
// definiteness test) are computed on the device; the host only chains the scalar comparisons in the reference's order.  One
// extra synchronisation per check (every 40 iterations) -- negligible next to the iterations it guards.
#include "device_utils.h"
#include <math.h>
#include <vector>

int32_t sync_ctl(cosmo_hip_handle* h);
int32_t psd_extreme_eigs(cosmo_hip_handle* h, const real* vec, real sign, real tol, std::vector<real>& lam_min);   // psd.hip

// delta_y capture at the top of the iteration: dy = rho .* (w_prev_s - s)            (solver.jl:145-148)
__global__ __launch_bounds__(COSMO_BS) void k_inf_capture(const Ctl* __restrict__ ctl, long long n, long long m,
                                                          const real* __restrict__ w_prev, const real* __restrict__ s,
                                                          const real* __restrict__ rho, real* __restrict__ dy) {
  if (ctl->halt) return;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS)
    dy[i] = rho[i] * (w_prev[n + i] - s[i]);
}

// dy -= mu_new ; dx = w_x - w_prev_x ; partials: max|E dy|, max|D dx|, sum q.dx           (solver.jl:331-335, infeasibility.jl:5,35,39)
__global__ __launch_bounds__(COSMO_BS) void k_inf_deltas(const Ctl* __restrict__ ctl, long long n, long long m, const real* __restrict__ w,
                                                         const real* __restrict__ w_prev, const real* __restrict__ s,
                                                         const real* __restrict__ rho, const real* __restrict__ E,
                                                         const real* __restrict__ Dd, const real* __restrict__ q,
                                                         real* __restrict__ dy, real* __restrict__ dx, real* __restrict__ p_ndy,
                                                         real* __restrict__ p_ndx, real* __restrict__ p_qdx) {
  if (ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  real ndy = 0.0, ndx = 0.0, qdx = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n + m; i += (long long)gridDim.x * COSMO_BS) {
    if (i < n) {
      const real d = w[i] - w_prev[i];
      dx[i] = d;
      ndx = amax(ndx, Dd[i] * d);
      qdx += q[i] * d;
    } else {
      const long long r = i - n;
      const real mu = rho[r] * (w_prev[i] - s[r]);
      const real d = dy[r] - mu;
      dy[r] = d;
      ndy = amax(ndy, E[r] * d);
    }
  }
  ndy = block_max(ndy, red); ndx = block_max(ndx, red); qdx = block_sum(qdx, red);
  if (threadIdx.x == 0) { p_ndy[blockIdx.x] = ndy; p_ndx[blockIdx.x] = ndx; p_qdx[blockIdx.x] = qdx; }
}

// one pass over [P | A'] with [dx; dy]: ||Dinv P dx||_inf and ||Dinv A' dy||_inf            (infeasibility.jl:12-17, 44-49)
__global__ __launch_bounds__(COSMO_BS) void k_inf_op(const Ctl* __restrict__ ctl, CsrView PT, const real* __restrict__ dx,
                                                     const real* __restrict__ dy, const real* __restrict__ Dinv,
                                                     real* __restrict__ p_pdx, real* __restrict__ p_ady) {
  if (ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real a = 0.0, b = 0.0;
  for (int k = blockIdx.x; k < PT.nb; k += gridDim.x)
    csr_stream_tile(PT, dx, dy, k, lds, red, [&](int row, real px, real aty) {
      const real d = Dinv[row];
      a = amax(a, px * d);
      b = amax(b, aty * d);
    });
  a = block_max(a, red); b = block_max(b, red);
  if (threadIdx.x == 0) { p_pdx[blockIdx.x] = a; p_ady[blockIdx.x] = b; }
}

// row-sharded runs: the same two norms with A' dy already summed over the ranks (aty) and P alone streamed
__global__ __launch_bounds__(COSMO_BS) void k_inf_op_rs(const Ctl* __restrict__ ctl, CsrView P, const real* __restrict__ dx,
                                                        const real* __restrict__ aty, const real* __restrict__ Dinv,
                                                        real* __restrict__ p_pdx, real* __restrict__ p_ady) {
  if (ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real a = 0.0, b = 0.0;
  for (int k = blockIdx.x; k < P.nb; k += gridDim.x)
    csr_stream_tile(P, dx, dx, k, lds, red, [&](int row, real s1, real s2) {
      const real d = Dinv[row];
      a = amax(a, (s1 + s2) * d);
      b = amax(b, aty[row] * d);
    });
  a = block_max(a, red); b = block_max(b, red);
  if (threadIdx.x == 0) { p_pdx[blockIdx.x] = a; p_ady[blockIdx.x] = b; }
}

// A dx scaled: adx = (Einv .* (A dx)) * (1/norm_dx)                                         (infeasibility.jl:53-59)
__global__ __launch_bounds__(COSMO_BS) void k_inf_adx(CsrView A, const real* __restrict__ dx, const real* __restrict__ Einv,
                                                      real inv_norm, real* __restrict__ adx) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  for (int k = blockIdx.x; k < A.nb; k += gridDim.x)
    csr_stream_tile(A, dx, dx, k, lds, red, [&](int row, real s1, real s2) { adx[row] = ((s1 + s2) * Einv[row]) * inv_norm; });
}

// primal certificate pieces on the simple rows: dyn = dy * (-1/norm) (in place) ; <dyn, b> ; Box support function ;
// in_dual(-dyn) violations of Nonnegatives rows.  meta: kind | boxindex << 2 (1x1 PSD rows are marked kind 2 as well).
__global__ __launch_bounds__(COSMO_BS) void k_inf_primal_rows(long long m, real fneg, real tol, const uint32_t* __restrict__ meta,
                                                              const real* __restrict__ bl, const real* __restrict__ bu,
                                                              const real* __restrict__ b, real* __restrict__ dy,
                                                              real* __restrict__ p_dtb, real* __restrict__ p_box, int* __restrict__ flags) {
  __shared__ real red[COSMO_BS / 64];
  real dtb = 0.0, box = 0.0;
  int viol = 0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS) {
    const real y = dy[i] * fneg;                    // delta_y *= (-1 / norm_dy)          (infeasibility.jl:19)
    dy[i] = y;
    dtb += y * b[i];
    const uint32_t mt = meta[i], kind = mt & 3u;
    if (kind == 3u) {                                 // Box support function (convexset.jl:850-856)
      const uint32_t j = mt >> 2;
      box += (fabs(y) > tol && y > R(0.0)) ? y * bu[j] : y * bl[j];
    } else if (kind == 2u) {                          // Nonnegatives: in_dual(-y): !any(x < -tol) (convexset.jl:76-78)
      if (-y < -tol) viol = 1;
    }
  }
  dtb = block_sum(dtb, red); box = block_sum(box, red);
  if (threadIdx.x == 0) { p_dtb[blockIdx.x] = dtb; p_box[blockIdx.x] = box; }
  if (viol) atomicOr(&flags[0], 1);
}

// dual certificate on the simple rows: in_pol_recc(adx)                                     (convexset.jl:34-36, 80-82, 859-861)
__global__ __launch_bounds__(COSMO_BS) void k_inf_dual_rows(long long m, real tol, const uint32_t* __restrict__ meta,
                                                            const real* __restrict__ bl,
                                                            const real* __restrict__ bu, const real* __restrict__ adx,
                                                            int* __restrict__ flags) {
  int viol = 0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS) {
    const real x = adx[i];
    const uint32_t mt = meta[i], kind = mt & 3u;
    if (kind == 1u) { if (fabs(x) > tol) viol = 1; }
    else if (kind == 2u) { if (x > tol) viol = 1; }
    else if (kind == 3u) {
      const uint32_t j = mt >> 2;
      if ((bu[j] == INFINITY && x > tol) || (bl[j] == -INFINITY && x < -tol)) viol = 1;
    }
  }
  if (viol) atomicOr(&flags[1], 1);
}

// SecondOrderCone membership, one wave per cone.  mode 0: in_dual(-y) : ||y[2:]|| <= tol + (-y[1])  (convexset.jl:116-118)
//                                                  mode 1: in_pol_recc(x): ||x[2:]|| <= tol - x[1]    (:120-122)
__global__ __launch_bounds__(COSMO_BS) void k_inf_soc(int ncones, const int* __restrict__ off, const int* __restrict__ dim,
                                                      const real* __restrict__ v, int mode, real tol, int* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * COSMO_BS + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * COSMO_BS) >> 6;
  for (int c = wave; c < ncones; c += nwaves) {
    const real* x = v + off[c];
    const int d = dim[c];
    if (d == 0) continue;
    real acc = 0.0;
    for (int i = 1 + lane; i < d; i += 64) { const real t = x[i]; acc += t * t; }
    const real nx = sqrt(wave_sum(acc));
    const real t0 = (mode == 0) ? -x[0] : x[0];
    const bool ok = (mode == 0) ? (nx <= tol + t0) : (nx <= tol - t0);
    if (!ok && lane == 0) atomicOr(&flags[mode], 1);
  }
}

static inline int ewg(long long N) {
  long long g = (N + COSMO_BS - 1) / COSMO_BS;
  if (g < 1) g = 1;
  if (g > COSMO_MAX_PARTIALS) g = COSMO_MAX_PARTIALS;
  return (int)g;
}
#define IPARTS(h, slot) ((h)->partials + (size_t)(slot) * COSMO_MAX_PARTIALS)

int32_t infeas_alloc(cosmo_hip_handle* h) {
  if (h->inf_dy) return COSMO_HIP_OK;
  HIPCHK(h, hipMalloc((void**)&h->inf_dy, sizeof(real) * (size_t)std::max<long long>(h->m, 1)));
  HIPCHK(h, hipMalloc((void**)&h->inf_dx, sizeof(real) * (size_t)std::max<long long>(h->n, 1)));
  HIPCHK(h, hipMalloc((void**)&h->inf_adx, sizeof(real) * (size_t)std::max<long long>(h->m, 1)));
  HIPCHK(h, hipMalloc((void**)&h->inf_flags, sizeof(int) * 4));
  HIPCHK(h, hipMemsetAsync(h->inf_dy, 0, sizeof(real) * (size_t)std::max<long long>(h->m, 1), h->stream));
  return COSMO_HIP_OK;
}

int32_t infeas_enqueue_capture(cosmo_hip_handle* h) {
  CHK(infeas_alloc(h));
  hipLaunchKernelGGL(k_inf_capture, dim3(ewg(h->m > 0 ? h->m : 1)), dim3(COSMO_BS), 0, h->stream, h->ctl, h->n, h->m, h->w_prev, h->s, h->rho,
                     h->inf_dy);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

static real host_max(const std::vector<real>& v, int n) { real a = 0.0; for (int i = 0; i < n; ++i) a = (v[i] > a || v[i] != v[i]) ? v[i] : a; return a; }
static real host_sum(const std::vector<real>& v, int n) { real a = 0.0; for (int i = 0; i < n; ++i) a += v[i]; return a; }

static int32_t fetch_parts(cosmo_hip_handle* h, int slot, int count, std::vector<real>& out) {
  out.resize(count);
  HIPCHK(h, hipMemcpyAsync(out.data(), IPARTS(h, slot), sizeof(real) * count, hipMemcpyDeviceToHost, h->stream));
  return COSMO_HIP_OK;
}

// The check of src/solver.jl:329-348, run after the iteration that followed a capture.  Sets *status to
// COSMO_HIP_PRIMAL_INFEASIBLE / COSMO_HIP_DUAL_INFEASIBLE or leaves it untouched.
int32_t infeas_check(cosmo_hip_handle* h, int32_t* status) {
  CHK(infeas_alloc(h));
  const long long n = h->n, m = h->m;
  const cosmo_hip_params& p = h->prm;
  const int gE = ewg(n + m);
  hipLaunchKernelGGL(k_inf_deltas, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, n, m, h->w, h->w_prev, h->s, h->rho, h->Escale, h->Dscale, h->q,
                     h->inf_dy, h->inf_dx, IPARTS(h, SLOT_AUX0), IPARTS(h, SLOT_AUX1), IPARTS(h, SLOT_AUX2));
  int gop = h->PT.grid;
  if (h->row_shard) {
    // A' dy = sum over the ranks of A_g' dy_g (one all-reduce of an n-vector per certificate test, i.e. every check_infeasibility iterations)
    gop = h->P.grid;
    CHK(launch_spmv_plain(h, h->AT, h->inf_dy, h->red_n));
    CHK(comm_allreduce_sum(h, h->red_n, (size_t)n));
    hipLaunchKernelGGL(k_inf_op_rs, dim3(gop), dim3(COSMO_BS), 0, h->stream, h->ctl, view_of(h->P), h->inf_dx, h->red_n, h->Dinv,
                       IPARTS(h, SLOT_RD), IPARTS(h, SLOT_MD));
  } else {
    hipLaunchKernelGGL(k_inf_op, dim3(h->PT.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, view_of(h->PT), h->inf_dx, h->inf_dy, h->Dinv,
                       IPARTS(h, SLOT_RD), IPARTS(h, SLOT_MD));
  }
  HIPCHK(h, hipGetLastError());
  std::vector<real> a0, a1, a2, b0, b1;
  CHK(fetch_parts(h, SLOT_AUX0, gE, a0)); CHK(fetch_parts(h, SLOT_AUX1, gE, a1)); CHK(fetch_parts(h, SLOT_AUX2, gE, a2));
  CHK(fetch_parts(h, SLOT_RD, gop, b0)); CHK(fetch_parts(h, SLOT_MD, gop, b1));
  CHK(sync_ctl(h));
  if (h->ctl_host->halt) return COSMO_HIP_OK;          // a status was decided earlier in the stream: nothing was computed
  const real epi = (real)p.eps_prim_inf, edi = (real)p.eps_dual_inf;      // settings are Float64 in the ABI, tests run in the model's type
  real norm_dy = host_max(a0, gE), norm_dx = host_max(a1, gE), q_dx = host_sum(a2, gE);
  real pdx_norm = host_max(b0, gop), ady_norm = host_max(b1, gop);
  if (h->row_shard) {
    // ||E dy||_inf is a max over all rows; the replicated scalars (their block partitions differ with m_loc, so q'dx may differ in the last
    // bits between ranks) are made IDENTICAL on every rank by the same reduction: the branches below must not diverge
    double v[5] = {(double)norm_dy, (double)norm_dx, (double)q_dx, (double)pdx_norm, (double)ady_norm};
    CHK(comm_allreduce_host(h, v, 5, 1));
    norm_dy = (real)v[0]; norm_dx = (real)v[1]; q_dx = (real)v[2]; pdx_norm = (real)v[3]; ady_norm = (real)v[4];
  }
  const ConeTable& C = h->cones;
  bool has_psd = false;
  for (size_t k = 0; k < C.type.size(); ++k)
    if ((C.type[k] == COSMO_HIP_PSD_SQUARE || C.type[k] == COSMO_HIP_PSD_TRIANGLE || C.type[k] == COSMO_HIP_PSD_TRIANGLE_COMPLEX) && C.dim[k] > 1) has_psd = true;
  // ---- is_primal_infeasible! (infeasibility.jl:1-29) ----
  if (norm_dy > epi && ady_norm <= epi * norm_dy) {
    HIPCHK(h, hipMemsetAsync(h->inf_flags, 0, sizeof(int) * 4, h->stream));
    const int gm = ewg(m > 0 ? m : 1);
    hipLaunchKernelGGL(k_inf_primal_rows, dim3(gm), dim3(COSMO_BS), 0, h->stream, m, -R(1.0) / norm_dy, epi, h->meta, h->box_l, h->box_u,
                       h->b, h->inf_dy, IPARTS(h, SLOT_AUX0), IPARTS(h, SLOT_AUX1), h->inf_flags);
    if (h->nsoc) hipLaunchKernelGGL(k_inf_soc, dim3(std::min(4096, (h->nsoc + 3) / 4)), dim3(COSMO_BS), 0, h->stream, h->nsoc, h->soc_off, h->soc_dim,
                                    h->inf_dy, 0, epi, h->inf_flags);
    CHK(cone3_enqueue_in_dual_neg(h, h->inf_dy, epi, h->inf_flags + 0));     // support_function!: in_dual(-dyn)
    HIPCHK(h, hipGetLastError());
    std::vector<real> d0, d1;
    int fl[4] = {0, 0, 0, 0};
    CHK(fetch_parts(h, SLOT_AUX0, gm, d0)); CHK(fetch_parts(h, SLOT_AUX1, gm, d1));
    HIPCHK(h, hipMemcpyAsync(fl, h->inf_flags, sizeof fl, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    bool in_dual_all = (fl[0] == 0);
    if (in_dual_all && has_psd) {
      // in_dual!(-dyn): is_pos_def(X + tol I)  <=>  lambda_min(X) > -tol with X = mat(-dyn)   (convexset.jl:415-418, algebra.jl:226-233)
      std::vector<real> lmin;
      CHK(psd_extreme_eigs(h, h->inf_dy, -1.0, epi, lmin));
      for (real l : lmin) if (!(l > -epi)) in_dual_all = false;
    }
    CHK(custom_test(h, h->inf_dy, 0, epi, &in_dual_all));               // user cones: in_dual(-dyn) through the callback
    if (h->comm) { int viol = in_dual_all ? 0 : 1; CHK(comm_allreduce_flag(h, &viol)); in_dual_all = (viol == 0); }   // owned cones only: combine
    real dyt_b = host_sum(d0, gm), box_sf = host_sum(d1, gm);
    if (h->row_shard) {                                   // <dyn, b> and the Box support function are sums over all rows
      double v[2] = {(double)dyt_b, (double)box_sf};
      CHK(comm_allreduce_host(h, v, 2, 0));
      dyt_b = (real)v[0]; box_sf = (real)v[1];
    }
    const real sF = (in_dual_all ? box_sf : INFINITY) - dyt_b;
    if (sF <= epi) { *status = COSMO_HIP_PRIMAL_INFEASIBLE; return COSMO_HIP_OK; }
  }
  // ---- is_dual_infeasible! (infeasibility.jl:32-68) ----
  const real c = R(1.0) / h->cinv;
  if (norm_dx > edi && q_dx / (norm_dx * c) < -edi && pdx_norm / (norm_dx * c) <= edi) {
    HIPCHK(h, hipMemsetAsync(h->inf_flags, 0, sizeof(int) * 4, h->stream));
    hipLaunchKernelGGL(k_inf_adx, dim3(h->A.grid > 0 ? h->A.grid : 1), dim3(COSMO_BS), 0, h->stream, view_of(h->A), h->inf_dx, h->Einv, R(1.0) / norm_dx,
                       h->inf_adx);
    hipLaunchKernelGGL(k_inf_dual_rows, dim3(ewg(m > 0 ? m : 1)), dim3(COSMO_BS), 0, h->stream, m, edi, h->meta,
                       h->box_l, h->box_u, h->inf_adx, h->inf_flags);
    if (h->nsoc) hipLaunchKernelGGL(k_inf_soc, dim3(std::min(4096, (h->nsoc + 3) / 4)), dim3(COSMO_BS), 0, h->stream, h->nsoc, h->soc_off, h->soc_dim,
                                    h->inf_adx, 1, edi, h->inf_flags);
    CHK(cone3_enqueue_in_dual_neg(h, h->inf_adx, edi, h->inf_flags + 1));    // in_pol_recc(v) = in_dual(-v)
    HIPCHK(h, hipGetLastError());
    int fl[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(fl, h->inf_flags, sizeof fl, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    bool in_recc = (fl[1] == 0);
    if (in_recc && has_psd) {
      // in_pol_recc!: is_neg_def(X, tol)  <=>  lambda_min(-X) > -tol                            (convexset.jl:421-424, algebra.jl:235-238)
      std::vector<real> lmin;
      CHK(psd_extreme_eigs(h, h->inf_adx, -1.0, edi, lmin));
      for (real l : lmin) if (!(l > -edi)) in_recc = false;
    }
    CHK(custom_test(h, h->inf_adx, 1, edi, &in_recc));
    if (h->comm) { int viol = in_recc ? 0 : 1; CHK(comm_allreduce_flag(h, &viol)); in_recc = (viol == 0); }
    if (in_recc) { *status = COSMO_HIP_DUAL_INFEASIBLE; return COSMO_HIP_OK; }
  }
  return COSMO_HIP_OK;
}
