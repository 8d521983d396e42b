// scaling.hip -- modified Ruiz equilibration of the device-resident problem (scale_ruiz!, src/scaling.jl:21-116).
// The reference runs it once per solve on the host over P, A, q, b and the Box bounds; here the same sequence runs on the
// device over the four resident CSR copies (A, A', P, [P | A']), so a caller can hand over the UNSCALED problem and never
// touch O(nnz) data on the host again.  Parity: column / row inf-norms are max-reductions (order free),
// the elementwise updates use the reference's expressions (nzval *= L[row] * R[col], src/algebra.jl:157-173), so D, E and
// the scaled data are bit-identical to a serial execution; only the cost scaling c depends on mean(col_norms(P)), whose
// summation order differs between Julia, NumPy and this file (pairwise, blocks of 128) at the last-ulp level.
#include "device_utils.h"
#include <math.h>
#include <algorithm>
#include <vector>

int32_t reclassify_after_scaling(cosmo_hip_handle* h);   // api.hip

namespace {

// v[col] = max(v[col], |val|) over every stored entry: non-negative doubles order like their bit patterns, so an integer
// atomic max is exact and order independent                                                     (col_norms!, src/algebra.jl:63-77)
__global__ __launch_bounds__(COSMO_BS) void k_col_absmax(long long nnz, const int* __restrict__ col, const real* __restrict__ val,
                                                         real* __restrict__ v) {
  for (long long k = (long long)blockIdx.x * COSMO_BS + threadIdx.x; k < nnz; k += (long long)gridDim.x * COSMO_BS)
#if REAL_IS_FLOAT
    atomicMax(reinterpret_cast<unsigned int*>(v + col[k]), __float_as_uint(fabs(val[k])));
#else
    atomicMax(reinterpret_cast<unsigned long long*>(v + col[k]), (unsigned long long)__double_as_longlong(fabs(val[k])));
#endif
}
// v[row] = max(v[row] (if !reset), max_k |val|) over the row                                  (row_norms!, src/algebra.jl:93-107)
__global__ __launch_bounds__(COSMO_BS) void k_row_absmax(int nrows, const int* __restrict__ rowptr, const real* __restrict__ val, int reset,
                                                         real* __restrict__ v) {
  for (int r = blockIdx.x * COSMO_BS + threadIdx.x; r < nrows; r += gridDim.x * COSMO_BS) {
    real a = reset ? 0.0 : v[r];
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) { const real t = fabs(val[k]); a = a > t ? a : t; }
    v[r] = a;
  }
}
// limit_scaling! then inv_sqrt!                                                                 (src/scaling.jl:10-13,125-127)
__global__ __launch_bounds__(COSMO_BS) void k_limit_inv_sqrt(long long n, real lo, real hi, real* __restrict__ v) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) {
    real s = v[i];
    s = s < lo ? 1.0 : (s > hi ? hi : s);          // clip(s, MIN_SCALING, MAX_SCALING, one(T))  (src/algebra.jl:5-7)
    v[i] = R(1.0) / sqrt(s);
  }
}
// nzval *= L[i] * R[j] for the entry (i, j) of the ORIGINAL matrix.  rows_are_cols: this CSR copy stores the transpose.
// L / R may be null (identity).  first_ncols: entries with col >= split_col belong to the second operand (merged [P | A']).
__global__ __launch_bounds__(COSMO_BS) void k_scale_csr(int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        real* __restrict__ val, const real* __restrict__ rowf,
                                                        const real* __restrict__ colf, int split_col, const real* __restrict__ colf2,
                                                        real cs1) {
  for (int r = blockIdx.x * COSMO_BS + threadIdx.x; r < nrows; r += gridDim.x * COSMO_BS) {
    const real fr = rowf ? rowf[r] : 1.0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int c = col[k];
      if (c < split_col) {
        real f = colf ? colf[c] : 1.0;
        real v = val[k] * (fr * f);
        if (cs1 != R(1.0)) v *= cs1;
        val[k] = v;
      } else {
        const real f = colf2 ? colf2[c - split_col] : 1.0;
        val[k] *= f * fr;
      }
    }
  }
}
__global__ __launch_bounds__(COSMO_BS) void k_scale_vals_below(int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                               real* __restrict__ val, int split_col, real cs) {
  for (int r = blockIdx.x * COSMO_BS + threadIdx.x; r < nrows; r += gridDim.x * COSMO_BS)
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) if (col[k] < split_col) val[k] *= cs;
}
__global__ __launch_bounds__(COSMO_BS) void k_vec_mul(long long n, real* __restrict__ x, const real* __restrict__ f) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) x[i] = f[i] * x[i];
}
__global__ __launch_bounds__(COSMO_BS) void k_vec_scal(long long n, real* __restrict__ x, real a) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) x[i] *= a;
}
__global__ __launch_bounds__(COSMO_BS) void k_vec_fill(long long n, real* __restrict__ x, real a) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) x[i] = a;
}
__global__ __launch_bounds__(COSMO_BS) void k_vec_recip(long long n, const real* __restrict__ x, real* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) y[i] = R(1.0) / x[i];
}
// Box bounds *= E on the Box rows (scale!(::Box), src/convexset.jl:863-867)
__global__ __launch_bounds__(COSMO_BS) void k_scale_box(long long m, const uint32_t* __restrict__ meta, const real* __restrict__ E,
                                                        real* __restrict__ bl, real* __restrict__ bu) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS) {
    const uint32_t mt = meta[i];
    if ((mt & 3u) == 3u) { const uint32_t j = mt >> 2; bl[j] *= E[i]; bu[j] *= E[i]; }
  }
}

inline int egrid(long long n) { long long g = (n + COSMO_BS - 1) / COSMO_BS; return (int)std::max<long long>(1, std::min<long long>(g, 4096)); }

real pairwise_sum(const real* v, size_t n) {
  if (n <= 128) { real s = 0.0; for (size_t i = 0; i < n; ++i) s += v[i]; return s; }
  const size_t h = n / 2;
  return pairwise_sum(v, h) + pairwise_sum(v + h, n - h);
}

}  // namespace

extern "C" int32_t cosmo_hip_scale_ruiz(cosmo_hip_handle* h, int64_t iterations, double min_scaling, double max_scaling, real* D_out,
                                        real* E_out, double* c_out) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (!h->have_problem || !h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "scale_ruiz: set_problem and set_cones first");
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "scale_ruiz: not available on a row-sharded handle (scale before cosmo_hip_set_row_shard: [P | A'] is gone and A is a row slice)");
  if (h->has_scaling) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "scale_ruiz: the problem already carries a scaling");
  if (!h->P_symmetric)
    return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "scale_ruiz: P must be structurally and numerically symmetric (the reference's symmetrize_full! is a host step)");
  if (iterations < 0 || !(min_scaling > 0.0) || !(max_scaling >= min_scaling)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "scale_ruiz: bad arguments");
  const long long n = h->n, m = h->m;
  hipStream_t st = h->stream;
  real *D = h->Dscale, *E = h->Escale, *Dw = h->Dinv, *Ew = h->Einv;    // the inverse scalings double as work vectors (scaling.jl:36-37)
  hipLaunchKernelGGL(k_vec_fill, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, D, 1.0);
  hipLaunchKernelGGL(k_vec_fill, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, E, 1.0);
  real c = 1.0;
  std::vector<real> hv((size_t)std::max<long long>(n, 1));
  const CsrDev &A = h->A, &AT = h->AT, &P = h->P, &PT = h->PT;
  auto scale_all = [&](const real* Dv, const real* Ev) {   // scale_data!(P, A, q, b, Ds, Es, 1)  (scaling.jl:157-168)
    if (Dv) hipLaunchKernelGGL(k_scale_csr, dim3(egrid(P.nrows)), dim3(COSMO_BS), 0, st, P.nrows, P.rowptr, P.col, P.val, Dv, Dv, INT32_MAX, (const real*)nullptr, 1.0);
    hipLaunchKernelGGL(k_scale_csr, dim3(egrid(A.nrows)), dim3(COSMO_BS), 0, st, A.nrows, A.rowptr, A.col, A.val, Ev, Dv, INT32_MAX, (const real*)nullptr, 1.0);
    hipLaunchKernelGGL(k_scale_csr, dim3(egrid(AT.nrows)), dim3(COSMO_BS), 0, st, AT.nrows, AT.rowptr, AT.col, AT.val, Dv, Ev, INT32_MAX, (const real*)nullptr, 1.0);
    // merged operator: row r, col < n -> P entry (D[r] D[c]); col >= n -> A' entry = A[c - n, r] (E[c - n] D[r])
    hipLaunchKernelGGL(k_scale_csr, dim3(egrid(PT.nrows)), dim3(COSMO_BS), 0, st, PT.nrows, PT.rowptr, PT.col, PT.val, Dv, Dv, (int)n, Ev, 1.0);
    if (Dv) hipLaunchKernelGGL(k_vec_mul, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, h->q, Dv);
    if (Ev) hipLaunchKernelGGL(k_vec_mul, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, h->b, Ev);
  };
  for (int64_t it = 0; it < iterations; ++it) {
    // kkt_col_norms! (scaling.jl:3-8): Dw = max(col norms P, col norms A) ; Ew = row norms A
    HIPCHK(h, hipMemsetAsync(Dw, 0, sizeof(real) * (size_t)std::max<long long>(n, 1), st));
    if (P.nnz) hipLaunchKernelGGL(k_col_absmax, dim3(egrid(P.nnz)), dim3(COSMO_BS), 0, st, P.nnz, P.col, P.val, Dw);
    hipLaunchKernelGGL(k_row_absmax, dim3(egrid(AT.nrows)), dim3(COSMO_BS), 0, st, AT.nrows, AT.rowptr, AT.val, 0, Dw);
    hipLaunchKernelGGL(k_row_absmax, dim3(egrid(A.nrows)), dim3(COSMO_BS), 0, st, A.nrows, A.rowptr, A.val, 1, Ew);
    hipLaunchKernelGGL(k_limit_inv_sqrt, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, min_scaling, max_scaling, Dw);
    hipLaunchKernelGGL(k_limit_inv_sqrt, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, min_scaling, max_scaling, Ew);
    scale_all(Dw, Ew);
    hipLaunchKernelGGL(k_vec_mul, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, D, Dw);      // lmul!(Dwork, D)
    hipLaunchKernelGGL(k_vec_mul, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, E, Ew);
    // cost scaling (scaling.jl:65-83): mean column norm of the scaled P and ||q||_inf
    HIPCHK(h, hipMemsetAsync(Dw, 0, sizeof(real) * (size_t)std::max<long long>(n, 1), st));
    if (P.nnz) hipLaunchKernelGGL(k_col_absmax, dim3(egrid(P.nnz)), dim3(COSMO_BS), 0, st, P.nnz, P.col, P.val, Dw);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(hv.data(), Dw, sizeof(real) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    const real mean_col_norm_P = n ? pairwise_sum(hv.data(), (size_t)n) / (real)n : 0.0;
    HIPCHK(h, hipMemcpyAsync(hv.data(), h->q, sizeof(real) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    real inf_norm_q = 0.0;
    for (long long i = 0; i < n; ++i) { const real a = fabs(hv[(size_t)i]); if (a > inf_norm_q || a != a) inf_norm_q = a; }
    if (mean_col_norm_P != R(0.0) && inf_norm_q != R(0.0)) {
      const real lo_ = (real)min_scaling, hi_ = (real)max_scaling;
      auto lim = [&](real s) { return s < lo_ ? R(1.0) : (s > hi_ ? hi_ : s); };
      inf_norm_q = lim(inf_norm_q);
      real scale_cost = std::max(inf_norm_q, mean_col_norm_P);
      scale_cost = lim(scale_cost);
      const real ctmp = R(1.0) / scale_cost;
      if (P.nnz) {
        hipLaunchKernelGGL(k_vec_scal, dim3(egrid(P.nnz)), dim3(COSMO_BS), 0, st, P.nnz, P.val, ctmp);       // scalarmul!(P, ctmp)
        hipLaunchKernelGGL(k_scale_vals_below, dim3(egrid(PT.nrows)), dim3(COSMO_BS), 0, st, PT.nrows, PT.rowptr, PT.col, PT.val, (int)n, ctmp);
      }
      hipLaunchKernelGGL(k_vec_scal, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, h->q, ctmp);
      c *= ctmp;
    }
  }
  // rectify_set_scalings! (scaling.jl:129-142): one scalar per SOC / PSD / exponential / power cone (convexset.jl:953-982)
  {
    const ConeTable& C = h->cones;
    bool changed = false;
    for (size_t k = 0; k < C.type.size(); ++k)
      if (C.type[k] >= COSMO_HIP_SOC && C.dim[k] > 0) changed = true;
    if (changed) {
      std::vector<real> Eh((size_t)m), Ewh((size_t)m, 1.0);
      HIPCHK(h, hipMemcpyAsync(Eh.data(), E, sizeof(real) * (size_t)m, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      for (size_t k = 0; k < C.type.size(); ++k) {
        if (C.type[k] < COSMO_HIP_SOC || C.dim[k] == 0) continue;
        const size_t o = (size_t)C.off[k], d = (size_t)C.dim[k];
        const real tmp = pairwise_sum(Eh.data() + o, d) / (real)d;       // rectify_scalar_scaling!: mean(E) ./ E
        for (size_t i = 0; i < d; ++i) Ewh[o + i] = tmp / Eh[o + i];
      }
      HIPCHK(h, hipMemcpyAsync(Ew, Ewh.data(), sizeof(real) * (size_t)m, hipMemcpyHostToDevice, st));
      HIPCHK(h, hipStreamSynchronize(st));
      scale_all(nullptr, Ew);                                                // scale_data!(P, A, q, b, I, Ework, 1)
      hipLaunchKernelGGL(k_vec_mul, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, E, Ew);
    }
  }
  // scale_sets! (scaling.jl:145-154): Box bounds
  if (h->cones.nbox_rows > 0) hipLaunchKernelGGL(k_scale_box, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, h->meta, E, h->box_l, h->box_u);
  // Dinv, Einv, c, cinv (scaling.jl:103-110)
  hipLaunchKernelGGL(k_vec_recip, dim3(egrid(n)), dim3(COSMO_BS), 0, st, n, D, h->Dinv);
  hipLaunchKernelGGL(k_vec_recip, dim3(egrid(m)), dim3(COSMO_BS), 0, st, m, E, h->Einv);
  HIPCHK(h, hipGetLastError());
  h->cinv = R(1.0) / c;
  h->has_scaling = true;
  if (D_out) { HIPCHK(h, hipMemcpyAsync(D_out, D, sizeof(real) * (size_t)n, hipMemcpyDeviceToHost, st)); }
  if (E_out) { HIPCHK(h, hipMemcpyAsync(E_out, E, sizeof(real) * (size_t)m, hipMemcpyDeviceToHost, st)); }
  HIPCHK(h, hipStreamSynchronize(st));
  if (c_out) *c_out = (double)c;
  // classify_constraints! runs on the SCALED b and bounds (setup.jl:36-37)
  return reclassify_after_scaling(h);
}
