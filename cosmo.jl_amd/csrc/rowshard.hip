// rowshard.hip -- row-sharded runs over the GPUs of one node (option 2, on top of a replicated n-side CG).
//
// The reference's loop couples the rows of a problem in exactly one place: the reduced KKT solve (src/solver.jl:50-55,
// src/linear_solver/kktsolver_indirect.jl:52-83).  Everything else of an iteration is row-local: the projections
// (src/convexset.jl:885-891: independent cones), the right-hand side ls_s = (b - 2 s) + w_s, nu = rho .* (A x_tl - ls_s), s_tl, the
// w_s update (src/solver.jl:50-65) and the primal residual (src/residuals.jl:2-9).  So rank g of N owns a contiguous range of CONES
// and with it
//     its rows of A          (h->A  : m_loc x n, for nu / s_tl / w_s and the primal residual),
//     its columns of A'      (h->AT : n x m_loc, for the partial products A_g' y_g),
//     its slices of b, rho, Einv, E, the row metadata, s, mu, s_tl, ls_s, nu and of the s-part of w / w_prev,
// while every n-vector (w_x, x_tl, the CG vectors) and the reduced operator of the CG are replicated and stay bit-identical on all
// ranks.  Per iteration the ranks exchange ONE all-reduce(sum) of the n-vector A'(rho .* ls_s) = sum_g A_g'(rho_g .* ls_s,g)
// (kktsolver_indirect.jl:52-54); per residual check one more all-reduce of A' mu with the 2 N per-rank primal norms appended
// (residuals.jl:12-18, 56-96) -- the north star's "all-reduce only".  The projected s is never exchanged.
//
// The CG itself (169 Krylov iterations per ADMM iteration on BASELINE config 5) runs redundantly on the ASSEMBLED / split reduced
// operator P + sigma I + A' rho A, which is built from the whole A before the conversion and refreshed from a replicated full-length
// rho (a function of the scalar ctl->rho and the row classes, both identical everywhere).  An all-reduce per operator application
// would cost more than the application (8.7 us on a 400 KB vector).
//
// cosmo_hip_set_row_shard converts a fully set-up handle (set_problem, set_cones, optional scale_ruiz, set_params, comm_init) in
// place: afterwards h->m is the LOCAL row count and every loop kernel runs unchanged on the local slices.
#include <algorithm>
#include <vector>
#include "device_utils.h"

int32_t rebuild_cone_plans(cosmo_hip_handle* h);                                                     // api.hip
int32_t comm_set_partition(cosmo_hip_handle* h, const int64_t* first_cone, const char* who);         // comm.hip
void comm_my_range(const cosmo_hip_handle* h, long long* cone_lo, long long* cone_hi, long long* row_lo, long long* row_hi);
int32_t launch_recover_mu(cosmo_hip_handle* h);                                                      // kernels.hip
bool aa_get_params(const cosmo_hip_handle* h, cosmo_hip_accel_params* out);                          // anderson.hip
void aa_free(cosmo_hip_handle* h);

template <class T>
static int32_t rs_alloc(cosmo_hip_handle* h, T** p, size_t count) {
  *p = nullptr;
  HIPCHK(h, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  HIPCHK(h, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
  return COSMO_HIP_OK;
}
template <class T>
static int32_t rs_fetch(cosmo_hip_handle* h, std::vector<T>& dst, const T* src, size_t count) {
  dst.resize(count);
  if (count == 0) return COSMO_HIP_OK;
  HIPCHK(h, hipMemcpyAsync(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}
// replace a global-length device vector by its [lo, lo + cnt) slice
template <class T>
static int32_t rs_slice(cosmo_hip_handle* h, T** vec, long long lo, long long cnt) {
  T* loc = nullptr;
  CHK(rs_alloc(h, &loc, (size_t)cnt));
  if (cnt > 0) HIPCHK(h, hipMemcpyAsync(loc, *vec + lo, sizeof(T) * (size_t)cnt, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  (void)hipFree(*vec);
  *vec = loc;
  return COSMO_HIP_OK;
}

void rs_free(cosmo_hip_handle* h) {
  if (h->rho_g) { (void)hipFree(h->rho_g); h->rho_g = nullptr; }
  if (h->rho_cls_g) { (void)hipFree(h->rho_cls_g); h->rho_cls_g = nullptr; }
  if (h->red_n) { (void)hipFree(h->red_n); h->red_n = nullptr; }
  h->row_shard = false;
}

// first_cone: nranks + 1 non-decreasing cone indices (as cosmo_hip_set_cone_shard).  Rank r owns the cones first_cone[r] <= k <
// first_cone[r+1] -- ALL kinds, also ZeroSet / Nonnegatives / Box -- and their rows.
extern "C" int32_t cosmo_hip_set_row_shard(cosmo_hip_handle* h, const int64_t* first_cone) {
  if (!h || !first_cone) return COSMO_HIP_ERR_INVALID;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (!h->have_params) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_row_shard: set_params first (the reduced operator is built from the whole A)");
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_row_shard: already row-sharded");
  if (h->prm.kkt_kind != COSMO_HIP_KKT_CG && h->prm.kkt_kind != COSMO_HIP_KKT_MINRES_REDUCED)
    return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_row_shard: the solvers of the REDUCED system only (kkt_kind CG / CG_SR / CG_JACOBI / MINRES_REDUCED); "
                      "MINRES on the full KKT system would need an all-reduce per operator application");
  cosmo_hip_accel_params accel_prm;
  const bool had_accel = aa_get_params(h, &accel_prm);      // the accelerator's history lives on w = [x ; rows]: re-created below on the local layout
  if (had_accel) aa_free(h);
  CHK(comm_set_partition(h, first_cone, "set_row_shard"));
  long long cone_lo, cone_hi, lo, hi;
  comm_my_range(h, &cone_lo, &cone_hi, &lo, &hi);
  const long long n = h->n, mg = h->m, ml = hi - lo;
  const int nranks = comm_nranks(h);

  // 1. the reduced operator must not depend on h->A / h->rho: force the split form (Am = rows with >= 2 nonzeros, diagonal from the
  //    singleton rows; assembled where sparse enough) and stop the single-launch CG, whose operands are sized at set_params time
  if (!h->op_split) { CHK(build_op_split(h, true)); CHK(choose_cg_recurrence(h)); }
  if (!h->op_split) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_row_shard: no split form of the reduced operator (empty A?)");
  if (h->pcg_on) { pcg_free(h); h->pcg_on = false; }

  // 2. local matrices from the device copies (scaled, if cosmo_hip_scale_ruiz ran)
  std::vector<int> arp, acol;
  std::vector<real> aval;
  CHK(rs_fetch(h, arp, (const int*)h->A.rowptr, (size_t)mg + 1));
  CHK(rs_fetch(h, acol, (const int*)h->A.col, (size_t)h->A.nnz));
  CHK(rs_fetch(h, aval, (const real*)h->A.val, (size_t)h->A.nnz));
  HostCsr Al, ATl;
  Al.nrows = (int)ml; Al.ncols = (int)n; Al.rowptr.assign((size_t)ml + 1, 0);
  const int z0 = arp[(size_t)lo], z1 = arp[(size_t)hi];
  for (long long i = 0; i <= ml; ++i) Al.rowptr[(size_t)i] = arp[(size_t)(lo + i)] - z0;
  Al.col.assign(acol.begin() + z0, acol.begin() + z1);
  Al.val.assign(aval.begin() + z0, aval.begin() + z1);
  ATl.nrows = (int)n; ATl.ncols = (int)ml; ATl.rowptr.assign((size_t)n + 1, 0);
  for (int c : Al.col) ATl.rowptr[(size_t)c + 1]++;
  for (long long j = 0; j < n; ++j) ATl.rowptr[(size_t)j + 1] += ATl.rowptr[(size_t)j];
  ATl.col.resize(Al.col.size()); ATl.val.resize(Al.col.size());
  { std::vector<int> pos(ATl.rowptr.begin(), ATl.rowptr.end() - 1);
    for (long long r = 0; r < ml; ++r)                               // rows ascending => every column of A_g keeps Julia's CSC order
      for (int k = Al.rowptr[(size_t)r]; k < Al.rowptr[(size_t)r + 1]; ++k) { const int p = pos[(size_t)Al.col[(size_t)k]]++; ATl.col[(size_t)p] = (int)r; ATl.val[(size_t)p] = Al.val[(size_t)k]; } }
  CHK(upload_csr(h, Al, h->A, (int)n));
  CHK(upload_csr(h, ATl, h->AT, (int)ml));
  free_csr(h->PT);                                                   // [P | A'] is replaced by P + the all-reduced A' product

  // 3. row data: the global rho / classes stay (the reduced operator is refreshed from them), everything else becomes a slice
  h->rho_g = h->rho; h->rho = nullptr;
  CHK(rs_alloc(h, &h->rho, (size_t)ml));
  if (ml > 0) HIPCHK(h, hipMemcpyAsync(h->rho, h->rho_g + lo, sizeof(real) * (size_t)ml, hipMemcpyDeviceToDevice, h->stream));
  h->rho_cls_g = h->rho_cls; h->rho_cls = nullptr;
  CHK(rs_alloc(h, &h->rho_cls, (size_t)ml));
  if (ml > 0) HIPCHK(h, hipMemcpyAsync(h->rho_cls, h->rho_cls_g + lo, sizeof(int) * (size_t)ml, hipMemcpyDeviceToDevice, h->stream));
  CHK(rs_slice(h, &h->b, lo, ml));
  CHK(rs_slice(h, &h->Einv, lo, ml));
  CHK(rs_slice(h, &h->Escale, lo, ml));
  CHK(rs_slice(h, &h->meta, lo, ml));                                // Box rows keep their index into the (whole) bound arrays
  if ((long long)h->rho_cls_host.size() == mg) h->rho_cls_host = std::vector<int32_t>(h->rho_cls_host.begin() + lo, h->rho_cls_host.begin() + hi);
  CHK(rs_alloc(h, &h->red_n, (size_t)(n + 2 * nranks + 8)));
  // the loop vectors (w, w_prev, s, mu, s_tl, ls_s, nu, y2, tmp_m, certificates) keep their allocations; only their first n + m_loc /
  // m_loc entries are used from here on, and set_iterates rewrites them

  // 4. this rank's composite set, offsets relative to its first row
  h->cones_g = h->cones;
  ConeTable L;
  for (long long k = cone_lo; k < cone_hi; ++k) {
    L.type.push_back(h->cones_g.type[(size_t)k]); L.dim.push_back(h->cones_g.dim[(size_t)k]); L.off.push_back(h->cones_g.off[(size_t)k] - lo);
    L.param.push_back(h->cones_g.param[(size_t)k]);
  }
  L.nbox_rows = h->cones_g.nbox_rows; L.box_l = h->cones_g.box_l; L.box_u = h->cones_g.box_u;
  h->cones = L;
  // user-defined cones (custom.hip): this rank keeps the callbacks of the cones it owns, with local cone indices / row offsets; the host staging
  // buffer keeps its size.  Certificates: custom_test sees the local slices, the verdicts are combined by comm_allreduce_flag (infeas.hip)
  { std::vector<CustomCone> mine;
    for (const CustomCone& cc : h->custom)
      if (cc.cone >= cone_lo && cc.cone < cone_hi) { CustomCone c2 = cc; c2.cone = cc.cone - cone_lo; c2.off = cc.off - lo; mine.push_back(c2); }
    h->custom = mine; }
  h->m_g = mg; h->row_lo = lo; h->m = ml;
  h->cone_lo = 0; h->cone_hi = -1;                                  // every local cone is owned
  h->row_shard = true;
  h->have_iterates = false;
  CHK(rebuild_cone_plans(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (had_accel) CHK(cosmo_hip_set_accelerator(h, &accel_prm));      // N = n + m_loc, inner products all-reduced (anderson.hip)
  return COSMO_HIP_OK;
}

// Every rank receives the GLOBAL iterates: w / w_prev = [x (replicated) ; s-part gathered], s, mu gathered (in-place all-gather of the
// ranks' row slices inside the staging buffer, which holds 2 (n + m_g) reals).
int32_t rs_get_iterates(cosmo_hip_handle* h, real* w, real* w_prev, real* s, real* mu) {
  const long long n = h->n, ml = h->m, mg = h->m_g, lo = h->row_lo;
  real* full = h->io;                                                // m_g reals
  auto gather = [&](const real* local_rows, real* host_out) -> int32_t {
    if (ml > 0) HIPCHK(h, hipMemcpyAsync(full + lo, local_rows, sizeof(real) * (size_t)ml, hipMemcpyDeviceToDevice, h->stream));
    CHK(comm_allgather_rows(h, full));
    if (mg > 0) HIPCHK(h, hipMemcpyAsync(host_out, full, sizeof(real) * (size_t)mg, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return COSMO_HIP_OK;
  };
  // all ranks pass the same set of non-null outputs (the collectives must match)
  if (mu) { CHK(launch_recover_mu(h)); CHK(gather(h->mu, mu)); }
  if (w) {
    if (n > 0) HIPCHK(h, hipMemcpyAsync(w, h->w, sizeof(real) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    CHK(gather(h->w + n, w + n));
  }
  if (w_prev) {
    if (n > 0) HIPCHK(h, hipMemcpyAsync(w_prev, h->w_prev, sizeof(real) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    CHK(gather(h->w_prev + n, w_prev + n));
  }
  if (s) CHK(gather(h->s, s));
  return COSMO_HIP_OK;
}

// out = {row_lo, row_hi, m_g, nnz(A_g), local cones, first local cone}
extern "C" int32_t cosmo_hip_row_shard_info(cosmo_hip_handle* h, int64_t out[6]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (!h->row_shard) { out[1] = h->m; out[2] = h->m; out[3] = h->A.nnz; out[4] = (int64_t)h->cones.type.size(); return COSMO_HIP_OK; }
  long long cone_lo, cone_hi, lo, hi;
  comm_my_range(h, &cone_lo, &cone_hi, &lo, &hi);
  out[0] = lo; out[1] = hi; out[2] = h->m_g; out[3] = h->A.nnz; out[4] = cone_hi - cone_lo; out[5] = cone_lo;
  return COSMO_HIP_OK;
}
