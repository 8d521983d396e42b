// api.hip -- C-ABI entry points of libcosmo_hip (include/cosmo_hip.h), handle lifecycle, host<->device staging of the
// problem (Julia CSC -> device CSR), and the speculative, device-controlled enqueue of the ADMM loop.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <chrono>
#include "internal.h"
#include "device_utils.h"

// launchers defined in kernels.hip
int32_t launch_project_simple_inplace(cosmo_hip_handle* h, real* s);
int32_t launch_z(cosmo_hip_handle* h, int guard);
int32_t launch_soc(cosmo_hip_handle* h, real* s, int guard);
int32_t launch_set_w(cosmo_hip_handle* h, const real* x0, const real* s0, const real* mu0);
int32_t launch_recover_mu(cosmo_hip_handle* h);
int32_t launch_rho_from_classes(cosmo_hip_handle* h, real rho0);
int32_t enqueue_cg_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count);
int32_t enqueue_cg_start(cosmo_hip_handle* h, int guard, real tol_k);
int32_t enqueue_rhs(cosmo_hip_handle* h, int guard);
int32_t enqueue_y2_only(cosmo_hip_handle* h);
int32_t enqueue_tail(cosmo_hip_handle* h, int loop_mode);
int32_t enqueue_count_solve(cosmo_hip_handle* h);
int32_t enqueue_clear_stall(cosmo_hip_handle* h);
int32_t enqueue_check(cosmo_hip_handle* h, int guard, int mode);
// minres.hip
int32_t minres_enqueue_solve(cosmo_hip_handle* h, int guard, bool from_loop);
int32_t minres_alloc(cosmo_hip_handle* h);
int32_t minres_resume(cosmo_hip_handle* h, int extra);

// ---------------------------------------------------------------------------------------------------------------------
int32_t cosmo_fail(cosmo_hip_handle* h, int32_t code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

template <class T>
static int32_t dalloc(cosmo_hip_handle* h, T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  HIPCHK(h, hipMalloc((void**)p, count * sizeof(T)));
  HIPCHK(h, hipMemsetAsync(*p, 0, count * sizeof(T), h->stream));
  return COSMO_HIP_OK;
}
template <class T>
static void dfree(T** p) { if (*p) { (void)hipFree(*p); *p = nullptr; } }

template <class T>
static int32_t h2d(cosmo_hip_handle* h, T* dst, const T* src, size_t count) {
  if (count == 0) return COSMO_HIP_OK;
  HIPCHK(h, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // host buffers are never retained after return
  return COSMO_HIP_OK;
}
template <class T>
static int32_t d2h(cosmo_hip_handle* h, T* dst, const T* src, size_t count) {
  if (count == 0) return COSMO_HIP_OK;
  HIPCHK(h, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}

// ---- profiling with HIP events on the handle's stream ----------------------------------------------------------------
void prof_begin(cosmo_hip_handle* h, int kc) {
  if (!h->profiling) return;
  if (h->ev_used + 2 > h->ev_pool.size()) {
    size_t old = h->ev_pool.size();
    h->ev_pool.resize(old + 1024);
    for (size_t i = old; i < h->ev_pool.size(); ++i) (void)hipEventCreate(&h->ev_pool[i]);
  }
  (void)hipEventRecord(h->ev_pool[h->ev_used], h->stream);
  h->ev_open.push_back(std::make_pair(kc, (int)h->ev_used));
  h->ev_used += 2;
}
void prof_end(cosmo_hip_handle* h) {
  if (!h->profiling) return;
  const int i = h->ev_open.back().second;
  (void)hipEventRecord(h->ev_pool[i + 1], h->stream);
}
int32_t prof_collect(cosmo_hip_handle* h) {
  if (!h->profiling) return COSMO_HIP_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (auto& pr : h->ev_open) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_pool[pr.second], h->ev_pool[pr.second + 1]) == hipSuccess) {
      h->kc_seconds[pr.first] += (double)ms * 1e-3;
      h->kc_launches[pr.first] += 1;
    }
  }
  h->ev_open.clear();
  h->ev_used = 0;
  return COSMO_HIP_OK;
}

// ---- CSR staging -----------------------------------------------------------------------------------------------------
void free_csr(CsrDev& D) {
  dfree(&D.rowptr); dfree(&D.col); dfree(&D.val); dfree(&D.split); dfree(&D.rb);
  D = CsrDev();
}

// Greedy CSR-stream schedule: consecutive rows whose nonzeros fit the LDS tile; a longer row gets its own block.
// Small matrices (the split CG operator of an SDP, small QPs) get smaller tiles so that a launch still has a few hundred
// workgroups: with the full 2048-nonzero tile a 100 k-nonzero operator is 50 workgroups on 256 CUs and its ~9 us are all ramp.
void build_row_blocks(const std::vector<int>& rowptr, int nrows, std::vector<int>& rb, int tile_override) {
  rb.clear();
  rb.push_back(0);
  const long long nnz_total = nrows > 0 ? (long long)rowptr[nrows] : 0;
  int tile = COSMO_NNZ_PER_BLOCK;
  if (nnz_total < 512LL * COSMO_NNZ_PER_BLOCK) tile = (int)std::max<long long>(256, ((nnz_total / 512 + 63) / 64) * 64);
  if (nnz_total <= 256LL * 768) tile = 256;     // operators of the single-launch CG (cg_persist.hip): one nonzero per thread and tile
  if (tile_override > 0) tile = std::min(tile_override, COSMO_NNZ_PER_BLOCK);
  const int ROWS_MAX = tile < COSMO_NNZ_PER_BLOCK ? COSMO_BS : 4 * COSMO_BS;
  int r = 0;
  while (r < nrows) {
    int r1 = r;
    long long cnt = 0;
    while (r1 < nrows) {
      const long long rn = (long long)rowptr[r1 + 1] - rowptr[r1];
      if (r1 > r && (cnt + rn > tile || r1 - r >= ROWS_MAX)) break;
      cnt += rn;
      ++r1;
      if (cnt > tile) break;  // single long row
    }
    rb.push_back(r1);
    r = r1;
  }
}

int32_t upload_csr(cosmo_hip_handle* h, const HostCsr& M, CsrDev& D, int split_col, int tile_override) {
  free_csr(D);
  D.nrows = M.nrows; D.ncols = M.ncols; D.nnz = (long long)M.val.size();
  D.split_col = split_col;
  std::vector<int> rbnd, rb;
  build_row_blocks(M.rowptr, M.nrows, rbnd, tile_override);
  D.nb = (int)rbnd.size() - 1;
  rb.resize((size_t)4 * std::max(D.nb, 1), 0);           // {r0, r1, nz0, nz1} per tile (16-byte aligned descriptors)
  for (int k = 0; k < D.nb; ++k) { rb[4 * k] = rbnd[k]; rb[4 * k + 1] = rbnd[k + 1]; rb[4 * k + 2] = M.rowptr[rbnd[k]]; rb[4 * k + 3] = M.rowptr[rbnd[k + 1]]; }
  int grid_cap = COSMO_MAX_PARTIALS;
  if (const char* e = getenv("COSMO_HIP_GRID_CAP")) { const int v = atoi(e); if (v >= 64 && v <= COSMO_MAX_PARTIALS) grid_cap = v; }   // lab knob
  D.grid = std::max(1, std::min(D.nb, grid_cap));
  D.xcd_affine = 1;
  if (const char* e = getenv("COSMO_HIP_XCD_AFFINE")) D.xcd_affine = atoi(e) ? 1 : 0;
  CHK(dalloc(h, &D.rowptr, (size_t)M.nrows + 1));
  CHK(dalloc(h, &D.col, M.col.size()));
  CHK(dalloc(h, &D.val, M.val.size()));
  CHK(dalloc(h, &D.rb, rb.size()));
  CHK(h2d(h, D.rowptr, M.rowptr.data(), M.rowptr.size()));
  CHK(h2d(h, D.col, M.col.data(), M.col.size()));
  CHK(h2d(h, D.val, M.val.data(), M.val.size()));
  CHK(h2d(h, D.rb, rb.data(), rb.size()));
  if (!M.split.empty()) {
    CHK(dalloc(h, &D.split, M.split.size()));
    CHK(h2d(h, D.split, M.split.data(), M.split.size()));
  }
  return COSMO_HIP_OK;
}

// Julia CSC (1-based Int64) of an (nr x nc) matrix -> CSR of the TRANSPOSE (free: same arrays) and CSR of the matrix.
static int32_t csc_to_csr_pair(cosmo_hip_handle* h, int64_t nr, int64_t nc, const int64_t* colptr, const int64_t* rowval,
                               const real* nzval, HostCsr& Mt, HostCsr& M) {
  const int64_t nnz = colptr[nc] - 1;
  if (colptr[0] != 1) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "colptr must be 1-based (colptr[0] == %lld)", (long long)colptr[0]);
  if (nnz < 0 || nnz >= (int64_t)2147483647) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "nnz out of int32 range");
  Mt.nrows = (int)nc; Mt.ncols = (int)nr;
  Mt.rowptr.resize(nc + 1); Mt.col.resize(nnz); Mt.val.resize(nnz);
  for (int64_t j = 0; j <= nc; ++j) {
    if (j > 0 && colptr[j] < colptr[j - 1]) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "colptr not monotone");
    Mt.rowptr[j] = (int)(colptr[j] - 1);
  }
  std::vector<int> cnt(nr + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t i = rowval[k] - 1;
    if (i < 0 || i >= nr) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "row index out of range");
    Mt.col[k] = (int)i; Mt.val[k] = nzval[k];
    cnt[i + 1]++;
  }
  M.nrows = (int)nr; M.ncols = (int)nc;
  M.rowptr.assign(nr + 1, 0);
  for (int64_t i = 0; i < nr; ++i) M.rowptr[i + 1] = M.rowptr[i] + cnt[i + 1];
  M.col.resize(nnz); M.val.resize(nnz);
  std::vector<int> pos(M.rowptr.begin(), M.rowptr.end() - 1);
  for (int64_t j = 0; j < nc; ++j)
    for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
      const int i = (int)(rowval[k] - 1);
      const int p = pos[i]++;
      M.col[p] = (int)j; M.val[p] = nzval[k];
    }
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int32_t cosmo_hip_version(void) { return COSMO_HIP_ABI_VERSION; }

extern "C" void cosmo_hip_default_params(cosmo_hip_params* p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->sigma = 1e-6; p->alpha = 1.6; p->rho = 0.1; p->eps_abs = 1e-5; p->eps_rel = 1e-5;
  p->eps_prim_inf = 1e-4; p->eps_dual_inf = 1e-4; p->tol_constant = 1.0; p->tol_exponent = 1.5;
  p->rho_min = 1e-6; p->rho_max = 1e6; p->rho_tol = 1e-4; p->rho_eq_over_rho_ineq = 1e3;
  p->adaptive_rho_tolerance = 5.0; p->cosmo_infty_min_scaling = 1e20 * 1e-4; p->time_limit = 0.0;
  p->max_iter = 5000; p->adaptive_rho_max_adaptions = INT64_MAX; p->kkt_kind = COSMO_HIP_KKT_CG;
  p->check_termination = 25; p->check_infeasibility = 40; p->adaptive_rho = 1; p->adaptive_rho_interval = 40;
  p->adaptive_rho_fraction = 0.4; p->setup_time = 0.0;
  p->unscale_residuals = 1;
  p->obj_true = (double)NAN; p->obj_true_tol = 1e-3;
}

extern "C" int32_t cosmo_hip_create(cosmo_hip_handle** out, int32_t device_id) {
  if (!out) return COSMO_HIP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return COSMO_HIP_ERR_HIP;  // no GPU: fail loudly, no CPU path
  if (device_id < 0 || device_id >= ndev) return COSMO_HIP_ERR_INVALID;
  cosmo_hip_handle* h = new cosmo_hip_handle();
  h->device = device_id;
  cosmo_hip_default_params(&h->prm);
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { delete h; return COSMO_HIP_ERR_HIP; }
  if (hipMalloc((void**)&h->ctl, sizeof(Ctl)) != hipSuccess || hipHostMalloc((void**)&h->ctl_host, sizeof(Ctl)) != hipSuccess ||
      hipMalloc((void**)&h->partials, sizeof(real) * COSMO_NSLOTS_TOTAL * COSMO_MAX_PARTIALS) != hipSuccess) {
    delete h; return COSMO_HIP_ERR_HIP;
  }
  (void)hipMemset(h->ctl, 0, sizeof(Ctl));
  (void)hipMemset(h->partials, 0, sizeof(real) * COSMO_NSLOTS_TOTAL * COSMO_MAX_PARTIALS);
  memset(h->ctl_host, 0, sizeof(Ctl));
  (void)hipEventCreate(&h->ev_proj0);
  (void)hipEventCreate(&h->ev_proj1);
  *out = h;
  return COSMO_HIP_OK;
}

static void free_vectors(cosmo_hip_handle* h) {
  dfree(&h->q); dfree(&h->b); dfree(&h->rho); dfree(&h->Dinv); dfree(&h->Einv); dfree(&h->Dscale); dfree(&h->Escale);
  dfree(&h->inf_dy); dfree(&h->inf_dx); dfree(&h->inf_adx); dfree(&h->inf_flags);
  dfree(&h->w); dfree(&h->w_prev); dfree(&h->s); dfree(&h->mu); dfree(&h->s_tl);
  dfree(&h->ls_x); dfree(&h->ls_s); dfree(&h->x_tl); dfree(&h->nu);
  dfree(&h->rhs); dfree(&h->r); dfree(&h->u); dfree(&h->c); dfree(&h->tmp_m); dfree(&h->y2); dfree(&h->mr); dfree(&h->cg_ru);
  dfree(&h->io);
}
static void free_cones(cosmo_hip_handle* h) {
  dfree(&h->meta); dfree(&h->box_l); dfree(&h->box_u); dfree(&h->rho_cls);
  dfree(&h->soc_off); dfree(&h->soc_dim); dfree(&h->soc_branch);
  psd_plan_destroy(h);
  cone3_free(h);
  custom_free(h);
  h->nsoc = 0;
}

extern "C" int32_t cosmo_hip_destroy(cosmo_hip_handle* h) {
  if (!h) return COSMO_HIP_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  free_csr(h->A); free_csr(h->AT); free_csr(h->P); free_csr(h->PT);
  free_op_split(h);
  pcg_free(h);
  sr_free(h);
  (void)cosmo_hip_comm_destroy(h);
  rs_free(h);
  aa_free(h);
  free_vectors(h);
  free_cones(h);
  dfree(&h->partials);
  if (h->ctl) { (void)hipFree(h->ctl); h->ctl = nullptr; }
  if (h->ctl_host) { (void)hipHostFree(h->ctl_host); h->ctl_host = nullptr; }
  for (auto& e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto& e : h->fb_ev) if (e) (void)hipEventDestroy(e);
  if (h->fb_k) (void)hipHostFree(h->fb_k);
  h->ev_pool.clear();
  if (h->ev_proj0) (void)hipEventDestroy(h->ev_proj0);
  if (h->ev_proj1) (void)hipEventDestroy(h->ev_proj1);
  if (h->stream) { (void)hipStreamDestroy(h->stream); h->stream = nullptr; }
  delete h;
  return COSMO_HIP_OK;
}

extern "C" const char* cosmo_hip_last_error(const cosmo_hip_handle* h) { return h ? h->err.c_str() : "null handle"; }

#define ENTER(h)                                                   \
  if (!(h)) return COSMO_HIP_ERR_INVALID;                          \
  if (hipSetDevice((h)->device) != hipSuccess) return cosmo_fail((h), COSMO_HIP_ERR_HIP, "hipSetDevice failed");

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int32_t cosmo_hip_set_problem(cosmo_hip_handle* h, int64_t n, int64_t m, const int64_t* P_colptr,
                                         const int64_t* P_rowval, const real* P_nzval, const int64_t* A_colptr,
                                         const int64_t* A_rowval, const real* A_nzval, const real* q, const real* b) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_problem: not available on a row-sharded handle");
  if (n < 0 || m < 0 || n + m >= 2147483647LL) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "n, m out of int32 range");
  if (!P_colptr || !A_colptr || (n > 0 && !q) || (m > 0 && !b)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "null pointer");
  h->n = n; h->m = m;
  HostCsr Pt, Pm, At, Am;
  CHK(csc_to_csr_pair(h, n, n, P_colptr, P_rowval, P_nzval, Pt, Pm));
  CHK(csc_to_csr_pair(h, m, n, A_colptr, A_rowval, A_nzval, At, Am));
  h->P_symmetric = (Pt.rowptr == Pm.rowptr && Pt.col == Pm.col && Pt.val == Pm.val);   // issymmetric(P) (scaling.jl:99)
  // row-merged operator [P | A'] : row j = (row j of P, columns < n) ++ (row j of A', columns shifted by n)
  HostCsr PT;
  PT.nrows = (int)n; PT.ncols = (int)(n + m);
  PT.rowptr.assign(n + 1, 0);
  PT.split.assign(n, 0);
  const size_t tot = Pm.val.size() + At.val.size();
  if (tot >= 2147483647ULL) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "nnz(P)+nnz(A) out of int32 range");
  PT.col.resize(tot); PT.val.resize(tot);
  size_t p = 0;
  for (int64_t j = 0; j < n; ++j) {
    PT.rowptr[j] = (int)p;
    for (int k = Pm.rowptr[j]; k < Pm.rowptr[j + 1]; ++k) { PT.col[p] = Pm.col[k]; PT.val[p] = Pm.val[k]; ++p; }
    PT.split[j] = (int)p;
    for (int k = At.rowptr[j]; k < At.rowptr[j + 1]; ++k) { PT.col[p] = At.col[k] + (int)n; PT.val[p] = At.val[k]; ++p; }
  }
  PT.rowptr[n] = (int)p;
  free_op_split(h);
  CHK(upload_csr(h, Am, h->A, (int)n));
  CHK(upload_csr(h, At, h->AT, (int)m));
  CHK(upload_csr(h, Pm, h->P, (int)n));
  CHK(upload_csr(h, PT, h->PT, (int)n));
  free_vectors(h);
  const size_t N = (size_t)(n + m);
  CHK(dalloc(h, &h->q, (size_t)n)); CHK(dalloc(h, &h->b, (size_t)m)); CHK(dalloc(h, &h->rho, (size_t)m));
  CHK(dalloc(h, &h->Dinv, (size_t)n)); CHK(dalloc(h, &h->Einv, (size_t)m)); CHK(dalloc(h, &h->Dscale, (size_t)n)); CHK(dalloc(h, &h->Escale, (size_t)m));
  { std::vector<real> ones((size_t)std::max<int64_t>(n, m), 1.0);
    CHK(h2d(h, h->Dinv, ones.data(), (size_t)n)); CHK(h2d(h, h->Einv, ones.data(), (size_t)m));
    CHK(h2d(h, h->Dscale, ones.data(), (size_t)n)); CHK(h2d(h, h->Escale, ones.data(), (size_t)m)); }
  CHK(dalloc(h, &h->w, N)); CHK(dalloc(h, &h->w_prev, N)); CHK(dalloc(h, &h->s, (size_t)m)); CHK(dalloc(h, &h->mu, (size_t)m));
  CHK(dalloc(h, &h->s_tl, (size_t)m)); CHK(dalloc(h, &h->ls_x, (size_t)n)); CHK(dalloc(h, &h->ls_s, (size_t)m));
  CHK(dalloc(h, &h->x_tl, (size_t)n)); CHK(dalloc(h, &h->nu, (size_t)m)); CHK(dalloc(h, &h->rhs, (size_t)n));
  CHK(dalloc(h, &h->r, (size_t)n)); CHK(dalloc(h, &h->u, (size_t)n)); CHK(dalloc(h, &h->c, (size_t)n));
  CHK(dalloc(h, &h->tmp_m, (size_t)m)); CHK(dalloc(h, &h->y2, (size_t)m)); CHK(dalloc(h, &h->io, 2 * N));
  CHK(h2d(h, h->q, q, (size_t)n));
  CHK(h2d(h, h->b, b, (size_t)m));
  h->has_scaling = false; h->cinv = 1.0;
  h->have_problem = true; h->have_cones = false; h->have_iterates = false;
  HIPCHK(h, hipMemsetAsync(h->ctl, 0, sizeof(Ctl), h->stream));
  h->host_iter = h->host_solves = 0; h->stalls = 0; h->budget = 12;
  return COSMO_HIP_OK;
}

// classify_constraints! (setup.jl:75-85, convexset.jl:62-69, 831-842) + apply_constraint_rho_scaling! classes
static int32_t classify_rows(cosmo_hip_handle* h, const std::vector<real>& bhost) {
  const ConeTable& C = h->cones;
  h->rho_cls_host.assign((size_t)h->m, 0);
  const real big = h->prm.cosmo_infty_min_scaling;
  size_t boxp = 0;
  for (size_t k = 0; k < C.type.size(); ++k) {
    const int64_t o = C.off[k], d = C.dim[k];
    if (C.type[k] == COSMO_HIP_ZERO) {
      for (int64_t i = 0; i < d; ++i) h->rho_cls_host[o + i] = 1;
    } else if (C.type[k] == COSMO_HIP_NONNEG) {
      for (int64_t i = 0; i < d; ++i) if (bhost[o + i] > big) h->rho_cls_host[o + i] = 2;
    } else if (C.type[k] == COSMO_HIP_BOX) {
      for (int64_t i = 0; i < d; ++i) {
        const real l = C.box_l[boxp + i], u = C.box_u[boxp + i];
        int c = 0;
        if (l < -big && u > big) c = 2;
        else if ((u - l) < (real)h->prm.rho_tol) c = 1;
        h->rho_cls_host[o + i] = c;
      }
      boxp += (size_t)d;
    }
  }
  CHK(dalloc(h, &h->rho_cls, (size_t)h->m));
  CHK(h2d(h, h->rho_cls, h->rho_cls_host.data(), (size_t)h->m));
  return COSMO_HIP_OK;
}

// after cosmo_hip_scale_ruiz: classify_constraints! sees the scaled b and Box bounds (setup.jl:36-37, 75-85)
int32_t reclassify_after_scaling(cosmo_hip_handle* h) {
  ConeTable& C = h->cones;
  if (C.nbox_rows > 0) {
    CHK(d2h(h, C.box_l.data(), h->box_l, (size_t)C.nbox_rows));
    CHK(d2h(h, C.box_u.data(), h->box_u, (size_t)C.nbox_rows));
  }
  std::vector<real> bhost((size_t)h->m);
  CHK(d2h(h, bhost.data(), h->b, (size_t)h->m));
  return classify_rows(h, bhost);
}

// SOC table and PSD plan of the cones this rank owns (all cones unless cosmo_hip_set_cone_shard restricted the range)
int32_t rebuild_cone_plans(cosmo_hip_handle* h) {
  const ConeTable& C = h->cones;
  std::vector<int> soc_off, soc_dim;
  h->soc_cone_index.clear();
  for (size_t k = 0; k < C.type.size(); ++k) {
    if (C.type[k] != COSMO_HIP_SOC || !cone_owned(h, (long long)k)) continue;
    soc_off.push_back((int)C.off[k]); soc_dim.push_back((int)C.dim[k]); h->soc_cone_index.push_back((int)k);
  }
  h->nsoc = (int)soc_off.size();
  CHK(dalloc(h, &h->soc_off, soc_off.size())); CHK(dalloc(h, &h->soc_dim, soc_dim.size())); CHK(dalloc(h, &h->soc_branch, soc_off.size()));
  if (h->nsoc) { CHK(h2d(h, h->soc_off, soc_off.data(), soc_off.size())); CHK(h2d(h, h->soc_dim, soc_dim.data(), soc_dim.size())); }
  CHK(cone3_plan_create(h));
  CHK(psd_plan_create(h));
  return COSMO_HIP_OK;
}

// How the PSD cones are projected (after set_cones; rebuilds the cone plans).  SIGN (default): side <= 16 by the wave-level Jacobi eigensolver, above it
// the verified matrix-sign iteration on the matrix cores (psd_polar.hip) -- the projection within its a-posteriori bound, nnz_lambda exact on gapped
// spectra.  EIGEN: the eigendecomposition-based projection the reference performs (src/convexset.jl:163-189, 243-263: syevr! + rank-k update) at EVERY
// side -- block one-sided Jacobi on G = X + ||X||_F I (psd.hip: one workgroup per cone up to side 256, multi-workgroup host-paced sweeps above), X+ =
// sum over lambda > 0 of lambda v v' and nnz_lambda counted from the eigenvalues themselves; several times slower on this chip (bench variants).
extern "C" int32_t cosmo_hip_set_psd_projection(cosmo_hip_handle* h, int32_t mode) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (mode != COSMO_HIP_PSD_PROJECTION_SIGN && mode != COSMO_HIP_PSD_PROJECTION_EIGEN) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_psd_projection: mode %d", (int)mode);
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (!h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_psd_projection: set_cones first");
  if (h->psd_mode == mode) return COSMO_HIP_OK;
  h->psd_mode = mode;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return rebuild_cone_plans(h);
}

extern "C" int32_t cosmo_hip_set_cones(cosmo_hip_handle* h, int64_t ncones, const int32_t* type, const int64_t* dim,
                                       const real* box_l, const real* box_u) {
  return cosmo_hip_set_cones_ex(h, ncones, type, dim, box_l, box_u, nullptr);
}

extern "C" int32_t cosmo_hip_set_cones_ex(cosmo_hip_handle* h, int64_t ncones, const int32_t* type, const int64_t* dim,
                                          const real* box_l, const real* box_u, const real* cone_param) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_cones: not available on a row-sharded handle");
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_problem must be called before set_cones");
  if (ncones < 0 || (ncones > 0 && (!type || !dim))) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "bad cone table");
  free_cones(h);
  ConeTable& C = h->cones;
  C = ConeTable();
  int64_t off = 0, nbox = 0;
  for (int64_t k = 0; k < ncones; ++k) {
    if (dim[k] < 0) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "negative cone dimension");
    if (type[k] < COSMO_HIP_ZERO || type[k] > COSMO_HIP_CUSTOM)
      return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "cone type %d is outside the hot-path scope", (int)type[k]);
    if (type[k] >= COSMO_HIP_EXP && type[k] <= COSMO_HIP_DUAL_POW && dim[k] != 3) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "exponential / power cones have dimension 3");
    if (type[k] == COSMO_HIP_POW || type[k] == COSMO_HIP_DUAL_POW) {
      if (!cone_param || !(cone_param[k] > R(0.0) && cone_param[k] < R(1.0)))
        return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "The exponent alpha of the power cone has to be in (0, 1).");
    }
    if (type[k] == COSMO_HIP_PSD_SQUARE || type[k] == COSMO_HIP_PSD_TRIANGLE_COMPLEX) {
      const int64_t r = (int64_t)llround(sqrt((real)dim[k]));
      if (r * r != dim[k]) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "PsdCone / complex PsdConeTriangle dimension must be a square");
    }
    C.type.push_back(type[k]); C.dim.push_back(dim[k]); C.off.push_back(off);
    C.param.push_back(cone_param ? cone_param[k] : 0.0);
    off += dim[k];
    if (type[k] == COSMO_HIP_BOX) nbox += dim[k];
  }
  if (off != h->m) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "cone dimensions sum to %lld but m = %lld", (long long)off, (long long)h->m);
  if (nbox > 0 && (!box_l || !box_u)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "Box cones need bounds");
  C.nbox_rows = nbox;
  if (nbox > 0) { C.box_l.assign(box_l, box_l + nbox); C.box_u.assign(box_u, box_u + nbox); }
  // per-row projection metadata (simple cones are projected by the elementwise copy kernel on every rank)
  std::vector<uint32_t> meta((size_t)h->m, 0u);
  int64_t boxp = 0;
  for (int64_t k = 0; k < ncones; ++k) {
    const int64_t o = C.off[k], d = C.dim[k];
    switch (C.type[k]) {
      case COSMO_HIP_ZERO: for (int64_t i = 0; i < d; ++i) meta[o + i] = 1u; break;
      case COSMO_HIP_NONNEG: for (int64_t i = 0; i < d; ++i) meta[o + i] = 2u; break;
      case COSMO_HIP_BOX:
        if (nbox >= (1LL << 30)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "too many Box rows");
        for (int64_t i = 0; i < d; ++i) meta[o + i] = 3u | ((uint32_t)(boxp + i) << 2);
        boxp += d;
        break;
      case COSMO_HIP_SOC: break;
      case COSMO_HIP_EXP: case COSMO_HIP_DUAL_EXP: case COSMO_HIP_POW: case COSMO_HIP_DUAL_POW: break;   // cone3.hip, in place
      case COSMO_HIP_CUSTOM: break;                                                                      // custom.hip, host callback
      case COSMO_HIP_PSD_SQUARE:
      case COSMO_HIP_PSD_TRIANGLE:
      case COSMO_HIP_PSD_TRIANGLE_COMPLEX:
        if (d == 1) for (int64_t i = 0; i < d; ++i) meta[o + i] = 2u;  // 1x1: max(x,0) (convexset.jl:307-308,404-405)
        break;
    }
  }
  CHK(dalloc(h, &h->meta, (size_t)h->m));
  CHK(h2d(h, h->meta, meta.data(), (size_t)h->m));
  CHK(dalloc(h, &h->box_l, (size_t)nbox)); CHK(dalloc(h, &h->box_u, (size_t)nbox));
  if (nbox > 0) { CHK(h2d(h, h->box_l, C.box_l.data(), (size_t)nbox)); CHK(h2d(h, h->box_u, C.box_u.data(), (size_t)nbox)); }
  h->cone_lo = 0; h->cone_hi = -1;
  CHK(rebuild_cone_plans(h));
  CHK(custom_plan_create(h));       // callbacks are (re)installed by cosmo_hip_set_custom_cone after every set_cones
  std::vector<real> bhost((size_t)h->m);
  CHK(d2h(h, bhost.data(), h->b, (size_t)h->m));
  CHK(classify_rows(h, bhost));
  h->have_cones = true;
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// CG operator split.  reduced_mul! applies  x -> P x + sigma x + A'(rho .* (A x))  (kktsolver_indirect.jl:57-64).  For a row i of
// A with exactly ONE nonzero a_i in column j (slack / identity / overlap rows: every PSD row of BASELINE configs 4 and 5), the
// contribution of that row to A' rho A is the scalar rho_i a_i^2 on the diagonal entry (j, j).  The CG iteration therefore only
// needs the rows with two or more nonzeros as a matrix (Am, compact) plus a diagonal that is rebuilt whenever rho changes:
//       x -> P x + (sigma + d) .* x + Am'(rho_m .* (Am x)),   d_j = sum_{single rows i, col j} rho_i a_i^2.
// Same linear operator, different association (rounding differs at the 1e-16 level; CG iteration counts within +-1 in the
// parity tests).  Enabled when the single rows hold at least half of nnz(A).  Residuals, rhs and tail keep the full matrices.
// Built from the DEVICE copies at set_params time, i.e. after an optional cosmo_hip_scale_ruiz.
// ---------------------------------------------------------------------------------------------------------------------
void free_op_split(cosmo_hip_handle* h) {
  free_csr(h->Am); free_csr(h->PTm);
  dfree(&h->op_mrow); dfree(&h->op_sc_ptr); dfree(&h->op_sc_row); dfree(&h->op_sc_a2); dfree(&h->op_diag); dfree(&h->op_rho_m);
  fold_free(h);
  h->op_split = false; h->op_nsingle = 0;
}

// force: build the split for ANY A (row-sharded runs need a reduced operator that does not depend on h->A, which becomes a row slice)
int32_t build_op_split(cosmo_hip_handle* h, bool force) {
  free_op_split(h);
  if (const char* e = getenv("COSMO_HIP_OP_SPLIT")) if (e[0] == '0' && !force) return COSMO_HIP_OK;
  // CG always; the reduced MINRES only when a row-sharded handle needs an operator that does not depend on the local slices (force)
  if (h->prm.kkt_kind != COSMO_HIP_KKT_CG && !(force && h->prm.kkt_kind == COSMO_HIP_KKT_MINRES_REDUCED)) return COSMO_HIP_OK;
  const long long n = h->n, m = h->m, nnzA = h->A.nnz, nnzP = h->P.nnz;
  if (m == 0 || nnzA == 0) return COSMO_HIP_OK;
  std::vector<int> arp((size_t)m + 1), acol((size_t)nnzA), prp((size_t)n + 1), pcol((size_t)std::max<long long>(nnzP, 1));
  std::vector<real> aval((size_t)nnzA), pval((size_t)std::max<long long>(nnzP, 1));
  CHK(d2h(h, arp.data(), h->A.rowptr, (size_t)m + 1)); CHK(d2h(h, acol.data(), h->A.col, (size_t)nnzA)); CHK(d2h(h, aval.data(), h->A.val, (size_t)nnzA));
  CHK(d2h(h, prp.data(), h->P.rowptr, (size_t)n + 1));
  if (nnzP) { CHK(d2h(h, pcol.data(), h->P.col, (size_t)nnzP)); CHK(d2h(h, pval.data(), h->P.val, (size_t)nnzP)); }
  long long nsingle = 0;
  for (long long i = 0; i < m; ++i) if (arp[i + 1] - arp[i] == 1) ++nsingle;
  if (nsingle * 2 < nnzA && !force) return COSMO_HIP_OK;
  // Am: rows with >= 2 nonzeros, compact
  HostCsr Am, AmT, PTm;
  std::vector<int> mrow;
  Am.ncols = (int)n; Am.rowptr.push_back(0);
  std::vector<int> sc_cnt((size_t)n + 1, 0);
  for (long long i = 0; i < m; ++i) {
    const int len = arp[i + 1] - arp[i];
    if (len >= 2) {
      mrow.push_back((int)i);
      for (int k = arp[i]; k < arp[i + 1]; ++k) { Am.col.push_back(acol[k]); Am.val.push_back(aval[k]); }
      Am.rowptr.push_back((int)Am.col.size());
    } else if (len == 1) {
      sc_cnt[(size_t)acol[arp[i]] + 1]++;
    }
  }
  Am.nrows = (int)mrow.size();
  const int mm = Am.nrows;
  // singles grouped by column, rows ascending (fixed summation order of the diagonal)
  std::vector<int> sc_ptr((size_t)n + 1, 0), sc_row((size_t)nsingle);
  std::vector<real> sc_a2((size_t)nsingle);
  for (long long j = 0; j < n; ++j) sc_ptr[j + 1] = sc_ptr[j] + sc_cnt[j + 1];
  { std::vector<int> pos(sc_ptr.begin(), sc_ptr.end() - 1);
    for (long long i = 0; i < m; ++i) if (arp[i + 1] - arp[i] == 1) { const int k = arp[i], j = acol[k], p = pos[j]++; sc_row[p] = (int)i; sc_a2[p] = aval[k] * aval[k]; } }
  // Am' (CSR over the n columns)
  AmT.nrows = (int)n; AmT.ncols = mm; AmT.rowptr.assign((size_t)n + 1, 0);
  for (int c : Am.col) AmT.rowptr[(size_t)c + 1]++;
  for (long long j = 0; j < n; ++j) AmT.rowptr[j + 1] += AmT.rowptr[j];
  AmT.col.resize(Am.col.size()); AmT.val.resize(Am.col.size());
  { std::vector<int> pos(AmT.rowptr.begin(), AmT.rowptr.end() - 1);
    for (int r = 0; r < mm; ++r) for (int k = Am.rowptr[r]; k < Am.rowptr[r + 1]; ++k) { const int p = pos[Am.col[k]]++; AmT.col[p] = r; AmT.val[p] = Am.val[k]; } }
  // merged [P | Am']
  PTm.nrows = (int)n; PTm.ncols = (int)(n + mm); PTm.rowptr.assign((size_t)n + 1, 0); PTm.split.assign((size_t)n, 0);
  PTm.col.reserve((size_t)nnzP + AmT.col.size()); PTm.val.reserve((size_t)nnzP + AmT.col.size());
  for (long long j = 0; j < n; ++j) {
    PTm.rowptr[j] = (int)PTm.col.size();
    for (int k = prp[j]; k < prp[j + 1]; ++k) { PTm.col.push_back(pcol[k]); PTm.val.push_back(pval[k]); }
    PTm.split[j] = (int)PTm.col.size();
    for (int k = AmT.rowptr[j]; k < AmT.rowptr[j + 1]; ++k) { PTm.col.push_back(AmT.col[k] + (int)n); PTm.val.push_back(AmT.val[k]); }
  }
  PTm.rowptr[n] = (int)PTm.col.size();
  CHK(upload_csr(h, Am, h->Am, (int)n));
  CHK(upload_csr(h, PTm, h->PTm, (int)n));
  CHK(dalloc(h, &h->op_mrow, (size_t)std::max(mm, 1))); CHK(dalloc(h, &h->op_sc_ptr, (size_t)n + 1)); CHK(dalloc(h, &h->op_sc_row, (size_t)nsingle));
  CHK(dalloc(h, &h->op_sc_a2, (size_t)nsingle)); CHK(dalloc(h, &h->op_diag, (size_t)n)); CHK(dalloc(h, &h->op_rho_m, (size_t)std::max(mm, 1)));
  if (mm) CHK(h2d(h, h->op_mrow, mrow.data(), (size_t)mm));
  CHK(h2d(h, h->op_sc_ptr, sc_ptr.data(), (size_t)n + 1)); CHK(h2d(h, h->op_sc_row, sc_row.data(), (size_t)nsingle)); CHK(h2d(h, h->op_sc_a2, sc_a2.data(), (size_t)nsingle));
  h->op_nsingle = nsingle;
  h->op_split = true;
  prp.resize((size_t)n + 1);
  CHK(fold_build(h, Am, prp, pcol, pval));      // assembled operator where Am' rho Am is sparse enough (cg_fold.hip)
  return refresh_op_split(h);
}

// LAB SWITCH (COSMO_HIP_CG_SR_DEFAULT=1; round 6): kkt_kind CG on an ASSEMBLED reduced operator runs the one-launch
// single-reduction recurrence (cg_sr.hip: k_sr_M) instead of the literal pair.  Measured as a candidate default and REJECTED (BASELINE config 5:
// 12.42 vs 11.50 us per Krylov iteration -- the 24-byte gathers of {r, w, s} from 32-byte records cost more than the launch they save, every
// XCD re-fetches the whole table through the fabric at each kernel boundary --, 236.6 vs 244.9 it/s; and at a 1e-10 stopping threshold the
// recurrence needs +1.2 % Krylov iterations, outside the +-1 per solve the parity tests allow; profiles/r06_cg_one_launch_default.txt).
int32_t choose_cg_recurrence(cosmo_hip_handle* h) {
  if (h->prm.kkt_kind != COSMO_HIP_KKT_CG || h->cg_jacobi || h->cg_sr || !h->op_fold) return COSMO_HIP_OK;
  if (((FoldPlan*)h->fold)->nd > 0) return COSMO_HIP_OK;          // a partially assembled operator belongs to the literal pair
  const char* e = getenv("COSMO_HIP_CG_SR_DEFAULT");
  if (!e || atoi(e) == 0) return COSMO_HIP_OK;
  h->cg_sr = true; h->cg_sr_auto = true;
  dfree(&h->cg_ru);                      // the {r, u} records of the literal pair
  return sr_alloc(h);
}

extern "C" int32_t cosmo_hip_set_params(cosmo_hip_handle* h, const cosmo_hip_params* p, const real* rho_vec) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_params: not available on a row-sharded handle");
  if (!p) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "null params");
  if (!h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_cones must be called before set_params");
  if (p->kkt_kind < COSMO_HIP_KKT_CG || p->kkt_kind > COSMO_HIP_KKT_CG_JACOBI) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "bad kkt_kind");
  if (p->check_termination <= 0) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "check_termination must be > 0");
  if (p->adaptive_rho && p->adaptive_rho_interval == 0 && !(p->adaptive_rho_fraction >= 0.0 && p->setup_time >= 0.0))
    return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "adaptive_rho_interval == 0 needs adaptive_rho_fraction >= 0 and setup_time >= 0");
  const bool reclass = (p->cosmo_infty_min_scaling != h->prm.cosmo_infty_min_scaling) || (p->rho_tol != h->prm.rho_tol);
  h->prm = *p;
  h->auto_rho_fixed_at = -1;
  h->cg_sr = (p->kkt_kind == COSMO_HIP_KKT_CG_SR);
  h->cg_sr_auto = false;
  h->cg_jacobi = (p->kkt_kind == COSMO_HIP_KKT_CG_JACOBI);
  if (h->cg_sr || h->cg_jacobi) h->prm.kkt_kind = COSMO_HIP_KKT_CG;      // the same reduced operator, split, budget and tail; only the Krylov recurrence differs
  if (reclass) {
    std::vector<real> bhost((size_t)h->m);
    CHK(d2h(h, bhost.data(), h->b, (size_t)h->m));
    CHK(classify_rows(h, bhost));
  }
  if (rho_vec) CHK(h2d(h, h->rho, rho_vec, (size_t)h->m));
  else CHK(launch_rho_from_classes(h, p->rho));
  // ws.rho / ws.rho_updates (set_rho_vec!, parameters.jl:3-13)
  CHK(d2h(h, h->ctl_host, h->ctl, 1));
  h->ctl_host->rho = p->rho;
  h->ctl_host->n_rho_updates = 1;
  h->ctl_host->rho_updates[0] = p->rho;
  h->ctl_host->solves = 0;
  h->ctl_host->kkt_iters_total = 0;
  CHK(h2d(h, h->ctl, h->ctl_host, 1));
  h->host_solves = 0;
  if (h->prm.kkt_kind != COSMO_HIP_KKT_CG) CHK(minres_alloc(h));
  CHK(sr_alloc(h));
  // the Krylov warm start (previous_solution) starts at zero (kktsolver_indirect.jl:32)
  HIPCHK(h, hipMemsetAsync(h->x_tl, 0, sizeof(real) * (size_t)std::max<long long>(h->n, 1), h->stream));
  HIPCHK(h, hipMemsetAsync(h->nu, 0, sizeof(real) * (size_t)std::max<long long>(h->m, 1), h->stream));
  h->have_params = true;
  // fused direction + A product (k_cg_dirA): one launch less per Krylov iteration, bit-identical; COSMO_HIP_CG_FUSE_DIR=0 disables it
  // (and with it the assembled operator of cg_fold.hip, which gathers the same {r, u} records)
  dfree(&h->cg_ru);
  { bool fuse = (h->prm.kkt_kind == COSMO_HIP_KKT_CG) && !h->cg_sr && h->n > 0;
    if (const char* e = getenv("COSMO_HIP_CG_FUSE_DIR")) fuse = fuse && atoi(e) != 0;
    if (fuse) CHK(dalloc(h, &h->cg_ru, 2 * (size_t)h->n)); }
  CHK(build_op_split(h));     // needs the (scaled) matrices and rho: both final from here on
  CHK(choose_cg_recurrence(h));
  if (h->cg_jacobi) {
    // the opt-in Jacobi-preconditioned CG lives on the ASSEMBLED reduced operator (its diagonal is the preconditioner): no silent fallback to the
    // unpreconditioned recurrence when the operator cannot be assembled (dense A' rho A: BASELINE config 2, where Jacobi makes the count worse anyway)
    if (!h->op_fold) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "kkt_kind CG_JACOBI needs the assembled reduced operator (csrc/cg_fold.hip: A' rho A too dense here, or COSMO_HIP_OP_FOLD=0 / COSMO_HIP_CG_FUSE_DIR=0)");
    // drop a persistent-CG state left by an earlier set_params with COSMO_HIP_CG_PERSIST=1: enqueue_solve_in_loop would otherwise run the
    // unpreconditioned persistent recurrence against the rebuilt operator
    h->pcg_on = false;
    if (h->pcg_sync) { (void)hipFree(h->pcg_sync); h->pcg_sync = nullptr; }
    if (h->pcg_u2) { (void)hipFree(h->pcg_u2); h->pcg_u2 = nullptr; }
    return COSMO_HIP_OK;
  }
  return pcg_setup(h);        // single-launch CG (opt-in)
}

extern "C" int32_t cosmo_hip_update_rho(cosmo_hip_handle* h, const real* rho_vec) {
  ENTER(h);
  h->fb_from = h->fb_recorded;               // (Krylov budget feedback: a new rho is a new regime)
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "update_rho: not available on a row-sharded handle");
  if (!h->have_params || !rho_vec) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "update_rho: not set up");
  CHK(h2d(h, h->rho, rho_vec, (size_t)h->m));
  return refresh_op_split(h);
}

static int32_t upload_or_ones(cosmo_hip_handle* h, real* dst, const real* src, size_t n) {
  if (src) return h2d(h, dst, src, n);
  std::vector<real> ones(n, 1.0);
  return h2d(h, dst, ones.data(), n);
}

extern "C" int32_t cosmo_hip_set_scaling_full(cosmo_hip_handle* h, const real* D, const real* Dinv, const real* E, const real* Einv,
                                              double c, double cinv) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_scaling: not available on a row-sharded handle");
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_problem first");
  CHK(upload_or_ones(h, h->Dinv, Dinv, (size_t)h->n)); CHK(upload_or_ones(h, h->Einv, Einv, (size_t)h->m));
  CHK(upload_or_ones(h, h->Dscale, D, (size_t)h->n)); CHK(upload_or_ones(h, h->Escale, E, (size_t)h->m));
  (void)c;
  h->cinv = cinv;
  h->has_scaling = true;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_set_scaling(cosmo_hip_handle* h, const real* Dinv, const real* Einv, double cinv) {
  ENTER(h);
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_problem first");
  // D and E (needed by the infeasibility tests only) are recovered as reciprocals
  std::vector<real> D, E;
  if (Dinv) { D.resize((size_t)h->n); for (long long i = 0; i < h->n; ++i) D[i] = R(1.0) / Dinv[i]; }
  if (Einv) { E.resize((size_t)h->m); for (long long i = 0; i < h->m; ++i) E[i] = R(1.0) / Einv[i]; }
  return cosmo_hip_set_scaling_full(h, Dinv ? D.data() : nullptr, Dinv, Einv ? E.data() : nullptr, Einv, 1.0 / cinv, cinv);
}

extern "C" int32_t cosmo_hip_update_qb(cosmo_hip_handle* h, const real* q, const real* b) {
  ENTER(h);
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_problem first");
  if (b && h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "update_qb(b): not available on a row-sharded handle");
  if (q) CHK(h2d(h, h->q, q, (size_t)h->n));
  if (b) {
    CHK(h2d(h, h->b, b, (size_t)h->m));
    if (h->have_cones) {  // setup! re-classifies on every optimize! (setup.jl:36-37)
      std::vector<real> bhost(b, b + h->m);
      CHK(classify_rows(h, bhost));
    }
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_rho_classes(cosmo_hip_handle* h, int32_t* cls) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "get_rho_classes: not available on a row-sharded handle");
  if (!h->have_cones || !cls) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "get_rho_classes: not set up");
  return d2h(h, cls, h->rho_cls, (size_t)h->m);
}
extern "C" int32_t cosmo_hip_get_rho_vec(cosmo_hip_handle* h, real* rho_vec) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "get_rho_vec: not available on a row-sharded handle");
  if (!h->have_params || !rho_vec) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "get_rho_vec: not set up");
  return d2h(h, rho_vec, h->rho, (size_t)h->m);
}

// ---------------------------------------------------------------------------------------------------------------------
int32_t sync_ctl(cosmo_hip_handle* h) {
  HIPCHK(h, hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  CHK(prof_collect(h));
  if (h->ctl_host->error) return cosmo_fail(h, h->ctl_host->error, "device raised error %d", h->ctl_host->error);
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_spmv(cosmo_hip_handle* h, int32_t which, real* y, const real* x) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "spmv: not available on a row-sharded handle");
  if (!h->have_problem || !x || !y) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "spmv: not set up");
  const CsrDev* M = which == COSMO_HIP_MAT_A ? &h->A : which == COSMO_HIP_MAT_AT ? &h->AT : which == COSMO_HIP_MAT_P ? &h->P : nullptr;
  if (!M) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "spmv: bad matrix id");
  // io holds n+m doubles: input first, output after it
  real* dx = h->io;
  real* dy = h->io + M->ncols;
  if ((long long)M->ncols + M->nrows > 2 * (h->n + h->m)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "spmv: staging too small");
  CHK(h2d(h, dx, x, (size_t)M->ncols));
  CHK(launch_spmv_plain(h, *M, dx, dy));
  CHK(d2h(h, y, dy, (size_t)M->nrows));
  h->spmv_calls[which] += 1;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_project(cosmo_hip_handle* h, real* s, int64_t* psd_rank_out, int32_t* soc_branch_out) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "project: not available on a row-sharded handle");
  if (!h->have_cones || (!s && h->m > 0)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "project: not set up");
  CHK(h2d(h, h->io, s, (size_t)h->m));
  CHK(launch_project_simple_inplace(h, h->io));
  CHK(launch_soc(h, h->io, 0));
  CHK(cone3_enqueue_project(h, h->io, 0));
  CHK(psd_enqueue_project(h, h->io, false));
  CHK(custom_enqueue_project(h, h->io, 0));
  CHK(comm_enqueue_exchange(h, h->io));
  CHK(d2h(h, s, h->io, (size_t)h->m));
  const size_t nc = h->cones.type.size();
  if (soc_branch_out) {
    for (size_t k = 0; k < nc; ++k) soc_branch_out[k] = -1;
    if (h->nsoc) {
      std::vector<int> br((size_t)h->nsoc);
      CHK(d2h(h, br.data(), h->soc_branch, (size_t)h->nsoc));
      for (int i = 0; i < h->nsoc; ++i) soc_branch_out[h->soc_cone_index[i]] = br[i];
    }
    CHK(cone3_get_branches(h, soc_branch_out));
  }
  if (psd_rank_out) {
    for (size_t k = 0; k < nc; ++k) psd_rank_out[k] = -1;
    CHK(psd_get_ranks(h, psd_rank_out));
    // 1x1 PSD cones are handled by the simple kernel
    for (size_t k = 0; k < nc; ++k)
      if ((h->cones.type[k] == COSMO_HIP_PSD_SQUARE || h->cones.type[k] == COSMO_HIP_PSD_TRIANGLE) && h->cones.dim[k] == 1)
        psd_rank_out[k] = s[h->cones.off[k]] > R(0.0) ? 1 : 0;
  }
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
static real tol_for_solve(const cosmo_hip_handle* h, long long k) {
  return h->prm.tol_constant / pow((real)k, h->prm.tol_exponent);  // get_tolerance, kktsolver_indirect.jl:168-170
}

static void adapt_budget(cosmo_hip_handle* h) {
  const int kmax = h->ctl_host->cg_k_max;
  int nb = kmax + 2;
  if (nb < 3) nb = 3;
  if (nb > 4096) nb = 4096;
  h->budget = nb;
}

__global__ void k_ctl_reset_kmax(Ctl* ctl) { ctl->cg_k_max = 0; }
__global__ void k_ctl_set_status(Ctl* ctl, int st) { ctl->status = st; ctl->halt = 1; }
static bool inf_due(const cosmo_hip_handle* h, long long it);

// after run_until reached an iteration with a pending infeasibility test: run it, publish a decided status
static int32_t maybe_infeas_check(cosmo_hip_handle* h, long long it) {
  if (h->ctl_host->status != 0 || !inf_due(h, it)) return COSMO_HIP_OK;
  int32_t st = 0;
  CHK(infeas_check(h, &st));
  if (st != 0) {
    hipLaunchKernelGGL(k_ctl_set_status, dim3(1), dim3(1), 0, h->stream, h->ctl, st);
    CHK(sync_ctl(h));
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_kkt_solve(cosmo_hip_handle* h, real* lhs, const real* rhs, int64_t* kkt_iters_out) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "kkt_solve: not available on a row-sharded handle");
  if (!h->have_params || !lhs || !rhs) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "kkt_solve: not set up");
  CHK(h2d(h, h->ls_x, rhs, (size_t)h->n));
  CHK(h2d(h, h->ls_s, rhs + h->n, (size_t)h->m));
  if (h->prm.kkt_kind == COSMO_HIP_KKT_CG) {
    CHK(enqueue_y2_only(h));
    CHK(enqueue_cg_start(h, 0, tol_for_solve(h, h->host_solves + 1)));
    int k = 0, chunk = std::max(h->budget, 4);
    bool solved = false;
    if (h->cg_sr) {
      CHK(sr_enqueue_start(h, 0));
      for (;;) {
        CHK(sr_enqueue_iterations(h, 0, k, chunk));
        CHK(sync_ctl(h));
        if (h->ctl_host->cg_done) break;
        k += chunk;
        chunk = std::min(chunk * 2, 1024);
      }
      solved = true;
    } else if (h->pcg_on) {
      CHK(pcg_enqueue_solve(h, 0));
      CHK(sync_ctl(h));
      if (h->ctl_host->stalled) {           // the start-up rendezvous failed: nothing was modified, continue with the multi-kernel loop
        h->pcg_on = false; h->pcg_fallbacks += 1;
        CHK(enqueue_clear_stall(h));
      } else {
        solved = true;
      }
    }
    while (!solved) {
      CHK(enqueue_cg_iterations(h, 0, k, chunk));
      CHK(sync_ctl(h));
      if (h->ctl_host->cg_done) break;
      k += chunk;
      chunk = std::min(chunk * 2, 1024);
    }
    CHK(enqueue_tail(h, 0));
    CHK(enqueue_count_solve(h));
  } else {
    CHK(minres_enqueue_solve(h, 0, false));
  }
  h->host_solves += 1;
  CHK(sync_ctl(h));
  if (kkt_iters_out) *kkt_iters_out = h->ctl_host->cg_k;
  CHK(d2h(h, lhs, h->x_tl, (size_t)h->n));
  CHK(d2h(h, lhs + h->n, h->nu, (size_t)h->m));
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_set_iterates(cosmo_hip_handle* h, const real* x0, const real* s0, const real* mu0) {
  ENTER(h);
  if (!h->have_params) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_iterates: set_params first");
  const long long n = h->n, m = h->m;
  // stage the three vectors in scratch buffers that the init step overwrites anyway (row-sharded: s0 / mu0 are the GLOBAL vectors, this
  // rank takes its slice)
  const real *dx = nullptr, *ds = nullptr, *dm = nullptr;
  if (h->row_shard) { if (s0) s0 += h->row_lo; if (mu0) mu0 += h->row_lo; }
  if (x0) { CHK(h2d(h, h->ls_x, x0, (size_t)n)); dx = h->ls_x; }
  if (s0) { CHK(h2d(h, h->ls_s, s0, (size_t)m)); ds = h->ls_s; }
  if (mu0) { CHK(h2d(h, h->tmp_m, mu0, (size_t)m)); dm = h->tmp_m; }
  CHK(launch_set_w(h, dx, ds, dm));
  HIPCHK(h, hipMemcpyAsync(h->w_prev, h->w, sizeof(real) * (size_t)(n + m), hipMemcpyDeviceToDevice, h->stream));
  // reset the loop counters (optimize! starts at iter = 0; the KKT solver's counters persist, setup.jl:54-61)
  CHK(d2h(h, h->ctl_host, h->ctl, 1));
  Ctl* c = h->ctl_host;
  c->halt = 0; c->status = 0; c->stalled = 0; c->error = 0; c->cg_done = 0; c->cg_k = 0; c->cg_k_max = 0; c->rho_changed = 0;
  c->iter = 0;
  c->r_prim = INFINITY; c->r_dual = INFINITY; c->max_norm_prim = 0; c->max_norm_dual = 0; c->cost = INFINITY;
  CHK(h2d(h, h->ctl, h->ctl_host, 1));
  h->host_iter = 0;
  h->have_iterates = true;
  h->fb_from = h->fb_recorded;               // Krylov counts of an earlier run say nothing about this one
  return COSMO_HIP_OK;
}

// ---- Krylov budget of a solve in the loop ---------------------------------------------------------------------------------------
// The loop enqueues a solve's Krylov iterations speculatively; what is enqueued beyond the iterations the solve needs are launches that
// return at once (4.6 + 1.1 us of device time per iteration all the same), and a budget that is too SMALL stalls the solve: everything
// enqueued behind it turns into no-ops and is enqueued again after the next synchronisation.  Until round 3 the budget was the largest
// count of the previous window + 2, fixed for a whole window: on BASELINE config 5 (bench window: iterations 11-50 in one call) the
// count creeps from 188 to 196, the window stalled twice and 65 % of its Krylov launches were no-ops; behind the rho update of
// iteration 40 the count drops to 115 -> 87 and the budget stayed at 200 (tools/speculation_waste.py).
// Now the count of every solve comes back through a pinned ring (4-byte copy + event behind k_tail), and solve s takes
//     budget = max + (max - min) / 2 + max / 40 + 3   of the counts K_{s-L-3} .. K_{s-L},      L = 2 solves of lag,
// after WAITING for the event of solve s - L (already complete unless the host is more than L iterations ahead), so that the budget is a
// function of the iteration history alone -- all ranks of a sharded run take the same decisions.  While the lagged solves predate a
// regime change (start of the loop, an adaptive-rho check, a stall) the window rule applies with 6 % of headroom.  Three stalls on feedback budgets switch
// the feedback off for the handle (COSMO_HIP_BUDGET_FEEDBACK=0 does so from the start).
static const int FB_LAG = 2;
static void feedback_reset(cosmo_hip_handle* h) { h->fb_from = h->fb_recorded; }
static int32_t solve_budget(cosmo_hip_handle* h, int* budget_out) {
  const int R = cosmo_hip_handle::FB_RING;
  bool& used = h->fb_used[(h->host_solves + 1) % R];         // this solve is number host_solves + 1 (ctl->solves counts completed ones)
  used = false;
  if (h->fb_mode == 1 && !h->fb_k) {
    if (const char* e = getenv("COSMO_HIP_BUDGET_FEEDBACK")) { if (atoi(e) == 0) h->fb_mode = 0; }
    if (h->fb_mode == 1 && hipHostMalloc((void**)&h->fb_k, sizeof(int) * cosmo_hip_handle::FB_RING) != hipSuccess) { (void)hipGetLastError(); h->fb_k = nullptr; h->fb_mode = 0; }
  }
  *budget_out = h->budget;
  if (h->fb_mode != 1 || h->profiling || h->exact_launches) return COSMO_HIP_OK;
  const long long j = h->fb_recorded - FB_LAG;
  const int wide = h->budget + h->budget / 16;        // regime change: the window rule with 6 % of headroom (a creeping count stalled it)
  if (j < h->fb_from || j < 0) { *budget_out = wide > 4096 ? 4096 : wide; return COSMO_HIP_OK; }
  // Bounded run-ahead: the host waits for the event of solve s - FB_LAG, i.e. it is never more than FB_LAG solves ahead of the device.  A
  // failure here is an error of the call, not a silent rank-local fallback: in a sharded run a different budget on one rank means
  // different stalls, a different number of re-enqueued collectives and in the end a hang inside RCCL instead of an error code.
  HIPCHK(h, hipEventSynchronize(h->fb_ev[j % R]));
  int kmax = h->fb_k[j % R], kmin = kmax, nv = 1;
  for (long long i = j - 1; i >= h->fb_from && i > j - 4; --i) { const int v = h->fb_k[i % R]; kmax = std::max(kmax, v); kmin = std::min(kmin, v); nv += 1; }
  // newest counts of the regime: their maximum + half their spread + 2.5 % + 3 (two or fewer counts: 15 % + 6 instead).  Replayed on the
  // recorded count sequences of BASELINE configs 2 and 5 (profiles/r03_cfg5_krylov_budget.txt): no stall, 6-10 % no-op iterations
  int b = (nv >= 3) ? kmax + (kmax - kmin) / 2 + kmax / 40 + 3 : kmax + (kmax * 15 + 99) / 100 + 6;
  if (b < 3) b = 3;
  if (b > 4096) b = 4096;
  used = true;
  h->cg_k_likely = kmax + 1;                  // iterations past the largest recent count: expected no-ops (check the flags first)
  *budget_out = b;
  return COSMO_HIP_OK;
}
static int32_t feedback_record(cosmo_hip_handle* h) {
  if (h->fb_mode != 1 || !h->fb_k) return COSMO_HIP_OK;
  const int R = cosmo_hip_handle::FB_RING;
  const int slot = (int)(h->fb_recorded % R);
  if (!h->fb_ev[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->fb_ev[slot], hipEventDisableTiming));
  HIPCHK(h, hipMemcpyAsync(&h->fb_k[slot], &h->ctl->cg_k, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(h->fb_ev[slot], h->stream));
  h->fb_recorded += 1;
  return COSMO_HIP_OK;
}

static int32_t enqueue_solve_in_loop(cosmo_hip_handle* h) {
  CHK(enqueue_rhs(h, 1));
  if (h->prm.kkt_kind == COSMO_HIP_KKT_CG) {
    CHK(enqueue_cg_start(h, 1, tol_for_solve(h, h->host_solves + 1)));
    if (h->cg_sr) {
      CHK(sr_enqueue_start(h, 1));
      if (h->exact_launches) {
        CHK(sr_enqueue_iterations(h, 1, 0, 0));
        for (int k = 0;; ++k) {
          CHK(sync_ctl(h));
          if (h->ctl_host->cg_done || h->ctl_host->halt) break;
          CHK(sr_enqueue_iterations(h, 1, k, 1));
        }
      } else {
        int bud = 0;
        CHK(solve_budget(h, &bud));
        CHK(sr_enqueue_iterations(h, 1, 0, bud));
        h->cg_k_likely = 0x7fffffff;
      }
    } else if (h->pcg_on) {
      CHK(pcg_enqueue_solve(h, 1));        // the whole Krylov loop in one launch (cg_persist.hip)
    } else if (h->exact_launches) {
      // measurement mode: one Krylov iteration per host round trip, so that every launch does full work
      CHK(enqueue_cg_iterations(h, 1, 0, 0));
      for (int k = 0;; ++k) {
        CHK(sync_ctl(h));
        if (h->ctl_host->cg_done || h->ctl_host->halt) break;
        CHK(enqueue_cg_iterations(h, 1, k, 1));
      }
    } else {
      int bud = 0;
      CHK(solve_budget(h, &bud));
      CHK(enqueue_cg_iterations(h, 1, 0, bud));
      h->cg_k_likely = 0x7fffffff;
    }
    CHK(enqueue_tail(h, 1));
    CHK(feedback_record(h));
  } else {
    CHK(minres_enqueue_solve(h, 1, true));
  }
  h->host_solves += 1;
  return COSMO_HIP_OK;
}

// One loop body (solver.jl:151-155) for iteration number `it` (1-based), enqueued without synchronisation.
// infeasibility schedule (solver.jl:326-349): the flag is set when it % check_infeasibility == 0 and consumed by the NEXT
// iteration: delta_y is captured at its top (:145-148) and the certificates are tested at its end.
static bool inf_due(const cosmo_hip_handle* h, long long it) {
  const long long ci = h->prm.check_infeasibility;
  if (ci <= 0 || ci > (1LL << 40)) return false;   // sharded runs: every rank tests its own cones, flags are max-reduced (comm.hip)
  return it > 1 && ((it - 1) % ci) == 0 && (it % ci) != 0;
}
static long long next_inf_iter(const cosmo_hip_handle* h, long long it) {   // smallest it' > it with inf_due(it'), or a huge value
  const long long ci = h->prm.check_infeasibility;
  if (ci <= 1 || ci > (1LL << 40)) return INT64_MAX;
  const long long k = (it <= 0) ? 1 : (it - 1) / ci + 1;   // it' = k ci + 1 with k >= 1 and it' > it
  return k * ci + 1;
}

static int32_t enqueue_iteration(cosmo_hip_handle* h, long long it) {
  if (inf_due(h, it)) CHK(infeas_enqueue_capture(h));
  CHK(launch_z(h, 1));
  CHK(launch_soc(h, h->s, 1));
  CHK(cone3_enqueue_project(h, h->s, 1));
  CHK(psd_enqueue_project(h, h->s, true));
  CHK(custom_enqueue_project(h, h->s, 1));
  CHK(comm_enqueue_exchange(h, h->s));      // clique sharding: the one exchange step of the iteration
  if (h->prm.adaptive_rho && h->prm.adaptive_rho_interval > 0 && (it % h->prm.adaptive_rho_interval) == 0) {
    CHK(enqueue_check(h, 1, 2));
    feedback_reset(h);                       // rho may change here (decided on the device): earlier Krylov counts say nothing about this solve
  }
  CHK(enqueue_solve_in_loop(h));
  h->host_iter = it;
  return COSMO_HIP_OK;
}

// If the Krylov budget of some iteration ran out, finish that iteration synchronously and re-sync the host counters.
static int32_t resolve_stall(cosmo_hip_handle* h) {
  while (h->ctl_host->stalled) {
    h->stalls += 1;
    // did the STALLED solve (number ctl->solves + 1: the counter advances when a solve completes) run on a feedback budget?  Solves
    // enqueued behind it were no-ops and say nothing; window-rule ('wide') budgets are not the feedback's stalls
    if (h->fb_mode == 1 && h->fb_used[(h->ctl_host->solves + 1) % cosmo_hip_handle::FB_RING]) { h->fb_stalls += 1; if (h->fb_stalls >= 3) h->fb_mode = 0; }
    feedback_reset(h);
    if (h->pcg_on) { h->pcg_on = false; h->pcg_fallbacks += 1; }     // only a failed start-up rendezvous stalls the persistent kernel
    int extra = std::max(2 * h->budget, 8);
    if (h->prm.kkt_kind == COSMO_HIP_KKT_CG) {
      CHK(enqueue_clear_stall(h));
      if (h->cg_sr) CHK(sr_enqueue_iterations(h, 1, h->ctl_host->cg_k, extra));
      else CHK(enqueue_cg_iterations(h, 1, h->ctl_host->cg_k, extra));
      CHK(enqueue_tail(h, 1));
    } else {
      CHK(enqueue_clear_stall(h));
      CHK(minres_resume(h, extra));
    }
    CHK(sync_ctl(h));
    h->budget = std::min(4096, std::max(h->budget, h->ctl_host->cg_k + 2));
  }
  h->host_iter = h->ctl_host->iter;
  h->host_solves = h->ctl_host->solves;
  return COSMO_HIP_OK;
}

// Enqueue iterations until the device has completed `target` iterations (or decided a status); optionally append
// a check (mode as enqueue_check) after the last one.  One FULL host synchronisation per call in the common case; with the budget
// feedback on, every solve also waits for the completion event of the solve FB_LAG = 2 before it (solve_budget), which bounds the
// host's run-ahead to two solves without draining the stream.
static int32_t run_until(cosmo_hip_handle* h, long long target, int check_mode) {
  for (;;) {
    for (long long it = h->host_iter + 1; it <= target; ++it) CHK(enqueue_iteration(h, it));
    if (check_mode >= 0) CHK(enqueue_check(h, 1, check_mode));
    CHK(sync_ctl(h));
    if (h->ctl_host->stalled) {
      CHK(resolve_stall(h));
      continue;  // re-enqueue what the stall skipped (later iterations and the check were no-ops)
    }
    adapt_budget(h);
    hipLaunchKernelGGL(k_ctl_reset_kmax, dim3(1), dim3(1), 0, h->stream, h->ctl);
    if (h->psd_polar) CHK(polar_adapt(h));
    return COSMO_HIP_OK;
  }
}

static int32_t admm_init_enqueue(cosmo_hip_handle* h) {
  // admm_x! ; admm_w! (solver.jl:137-138).  The device iteration counter must not advance: compensate afterwards.
  CHK(enqueue_solve_in_loop(h));
  CHK(sync_ctl(h));
  if (h->ctl_host->stalled) {
    long long keep_iter = 0;
    CHK(resolve_stall(h));
    (void)keep_iter;
  }
  // undo the iteration count of the init step
  CHK(d2h(h, h->ctl_host, h->ctl, 1));
  h->ctl_host->iter = 0;
  h->ctl_host->cg_k_max = std::max(h->ctl_host->cg_k_max, h->ctl_host->cg_k);
  CHK(h2d(h, h->ctl, h->ctl_host, 1));
  adapt_budget(h);
  h->host_iter = 0;
  h->host_solves = h->ctl_host->solves;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_admm_init(cosmo_hip_handle* h) {
  ENTER(h);
  if (!h->have_iterates) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "admm_init: set_iterates first");
  return admm_init_enqueue(h);
}

extern "C" int32_t cosmo_hip_admm_iterate(cosmo_hip_handle* h, int64_t n_iters) {
  ENTER(h);
  if (!h->have_iterates) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "admm_iterate: set_iterates first");
  const long long target = h->host_iter + n_iters;
  const long long CHUNK = 50;
  while (h->host_iter < target) {
    const long long t = std::min(target, h->host_iter + CHUNK);
    CHK(run_until(h, t, -1));
    if (h->ctl_host->status != 0) break;
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_admm_iterate_checked(cosmo_hip_handle* h, int64_t n_iters, int32_t* status_out) {
  ENTER(h);
  if (!h->have_iterates) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "admm_iterate_checked: set_iterates first");
  const long long target = h->host_iter + n_iters;
  const long long ct = h->prm.check_termination;
  if (status_out) *status_out = COSMO_HIP_UNDETERMINED;
  while (h->host_iter < target) {
    const long long it = h->host_iter;
    long long next = (it == 0) ? 1 : ((it / ct) + 1) * ct;
    next = std::min(next, next_inf_iter(h, it));
    if (next > target) next = target;
    const bool check = (next % ct == 0) || next == 1;
    CHK(run_until(h, next, check ? 1 : -1));
    CHK(maybe_infeas_check(h, next));
    if (h->ctl_host->status != 0) { if (status_out) *status_out = h->ctl_host->status; break; }
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_residuals(cosmo_hip_handle* h, double out[5]) {
  ENTER(h);
  if (!h->have_iterates || !out) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "residuals: set_iterates first");
  CHK(enqueue_check(h, 0, 3));
  CHK(sync_ctl(h));
  const Ctl* c = h->ctl_host;
  out[0] = (double)c->r_prim; out[1] = (double)c->r_dual; out[2] = (double)c->max_norm_prim; out[3] = (double)c->max_norm_dual; out[4] = (double)c->cost;
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The accelerated loop (src/solver.jl:140-165 with acceleration_pre! / acceleration_post!, src/accelerator_interface.jl).
// One host synchronisation per iteration (two when a deferred rho update / infeasibility check is pending or the
// safeguard has to be evaluated): the success flag of the least-squares step and the safeguarding decision are taken on
// the device and steer what the host enqueues next, exactly where the reference branches on CA.was_successful.
// ---------------------------------------------------------------------------------------------------------------------
static int32_t enqueue_admm_step(cosmo_hip_handle* h, bool rho_rules) {
  CHK(launch_z(h, 1));
  CHK(launch_soc(h, h->s, 1));
  CHK(cone3_enqueue_project(h, h->s, 1));
  CHK(psd_enqueue_project(h, h->s, true));
  CHK(custom_enqueue_project(h, h->s, 1));
  CHK(comm_enqueue_exchange(h, h->s));      // clique sharding: the one exchange step of the iteration (row-sharded handles: no-op)
  if (rho_rules) CHK(enqueue_check(h, 1, 2));
  CHK(enqueue_solve_in_loop(h));
  return COSMO_HIP_OK;
}
static int32_t sync_and_resolve(cosmo_hip_handle* h) {
  CHK(sync_ctl(h));
  if (h->ctl_host->stalled) CHK(resolve_stall(h));
  adapt_budget(h);
  hipLaunchKernelGGL(k_ctl_reset_kmax, dim3(1), dim3(1), 0, h->stream, h->ctl);
  return COSMO_HIP_OK;
}

// The reference tests the wall clock (solver.jl:351-354).  In a sharded run the ranks' clocks differ, their decisions must not: every time-limit
// decision is taken on the MAXIMUM of the ranks' elapsed seconds (one tiny host all-reduce through the communicator; identical on all ranks).
static int32_t collective_elapsed(cosmo_hip_handle* h, const std::chrono::steady_clock::time_point t0, double* el) {
  *el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (h->comm && comm_nranks(h) > 1) CHK(comm_allreduce_host(h, el, 1, 1));
  return COSMO_HIP_OK;
}

// The reference's automatic rho interval (apply_rho_adaptation_rules!, solver.jl:244-256): `settings.adaptive_rho_interval == 0` means "fix it once the
// loop has run for adaptive_rho_fraction * setup_time seconds" -- to round_multiple(iter, check_termination) (algebra.jl:245-247), at least
// check_termination (25 where that is 0).  Called where the host knows that the device has finished iteration `it` (at the top of iteration it + 1 in
// the reference's terms).  The rule fires once; the interval then lives in h->prm like a fixed one, and what the device does from there on is the
// fixed-interval schedule.  Sharded runs: BOTH sides of the comparison are rank-local (every process measures its own loop time and its own
// setup_time -- model.optimize() times its own setup_row_sharding), so the DECISION is made collective: the ranks all-reduce (max) the margin
// elapsed - fraction * setup_time and fire together as soon as any of them would.  A rank that fired alone would stop calling this all-reduce while its
// peers go on, and schedule its rho checks (and their all-reduces) at other iterations: mismatched collectives, i.e. a hang inside RCCL.
static int32_t auto_rho_interval(cosmo_hip_handle* h, long long iter, const std::chrono::steady_clock::time_point t0) {
  cosmo_hip_params& p = h->prm;
  if (!(p.adaptive_rho && p.adaptive_rho_interval == 0)) return COSMO_HIP_OK;
  double margin = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - p.adaptive_rho_fraction * p.setup_time;
  if (h->comm && comm_nranks(h) > 1) CHK(comm_allreduce_host(h, &margin, 1, 1));
  if (!(margin > 0.0)) return COSMO_HIP_OK;
  const long long N = p.check_termination > 0 ? p.check_termination : 25;
  const double x = (double)iter + 0.5 * (double)N;
  long long v = (long long)floor(x - fmod(x, (double)N));                 // round_multiple(iter, N)
  v = std::max(v, N);
  p.adaptive_rho_interval = (int32_t)std::min<long long>(v, INT32_MAX);
  h->auto_rho_fixed_at = iter;
  return COSMO_HIP_OK;
}

static int32_t optimize_accelerated(cosmo_hip_handle* h, int* status_out, long long* iter_out, const std::chrono::steady_clock::time_point t0) {
  const cosmo_hip_params& p = h->prm;
  // Sharded runs: with clique (cone) sharding every rank holds the whole w and all scalars are computed redundantly and bit-identically, so the
  // accelerator needs nothing; with row sharding its inner products are all-reduced (anderson.hip).  Either way every rank takes the same
  // branches below (success / declined flags come from identical scalars).
  CHK(aa_begin_solve(h));
  h->safeguarding_iter = 0;
  CHK(admm_init_enqueue(h));
  int status = COSMO_HIP_UNDETERMINED;
  long long it = 0;
  bool inf_check_due = false, rho_update_due = false;
  const long long ct = p.check_termination, ci = p.check_infeasibility;
  int n_rho_seen = h->ctl_host->n_rho_updates;
  while (it + h->safeguarding_iter < p.max_iter) {
    it += 1;
    bool attempted = false;
    int success = 0, declined = 0;
    CHK(aa_enqueue_pre(h, it, &attempted));                                   // acceleration_pre!
    CHK(auto_rho_interval(h, it, t0));                                        // solver.jl:244-256 (this loop synchronises every iteration: the reference's own test point)
    if (p.adaptive_rho && p.adaptive_rho_interval > 0 && (it % p.adaptive_rho_interval) == 0 &&
        (long long)(n_rho_seen - 1) < p.adaptive_rho_max_adaptions)
      rho_update_due = true;                                                  // solver.jl:262-264
    bool have_success = !attempted;
    if (attempted && (inf_check_due || rho_update_due)) { CHK(aa_fetch_flags(h, &success, nullptr)); have_success = true; }
    if (inf_check_due && have_success && !success) CHK(infeas_enqueue_capture(h));      // solver.jl:145-148
    const bool do_rho = rho_update_due && have_success && !success;          // update_suggested (solver.jl:268,284-292)
    if (do_rho) rho_update_due = false;
    CHK(enqueue_admm_step(h, do_rho));
    CHK(sync_and_resolve(h));
    if (h->ctl_host->error) return cosmo_fail(h, h->ctl_host->error, "device error %d in the accelerated loop", h->ctl_host->error);
    if (!have_success) CHK(aa_fetch_flags(h, &success, nullptr));
    if (do_rho && h->ctl_host->n_rho_updates != n_rho_seen) {                 // solver.jl:272-275
      n_rho_seen = h->ctl_host->n_rho_updates;
      CHK(aa_restart(h));
      aa_note_rho_restart(h);
    }
    if (aa_active(h) && success) {                                            // acceleration_post!
      if (aa_safeguarded(h)) {
        CHK(aa_enqueue_guard(h));
        CHK(aa_fetch_flags(h, nullptr, &declined));
        if (declined) {
          CHK(aa_enqueue_reset(h));
          CHK(enqueue_admm_step(h, false));
          CHK(sync_and_resolve(h));
          h->safeguarding_iter += 1;
        }
      }
      aa_count(h, 1, declined);
    }
    if ((it % ct) == 0 || it == 1) {                                          // check_termination! (solver.jl:306-323)
      CHK(enqueue_check(h, 1, 1));
      CHK(sync_ctl(h));
      aa_check_accuracy_activation(h, h->ctl_host->r_prim, h->ctl_host->r_dual, h->ctl_host->max_norm_prim, h->ctl_host->max_norm_dual);
      if (h->ctl_host->status != 0) { status = h->ctl_host->status; break; }
    }
    if (ci > 0 && ci < (1LL << 40) && (it % ci) == 0) {
      inf_check_due = true;
    } else if (inf_check_due && !success) {                                   // solver.jl:329-348
      inf_check_due = false;
      int32_t st = 0;
      CHK(infeas_check(h, &st));
      if (st != 0) {
        hipLaunchKernelGGL(k_ctl_set_status, dim3(1), dim3(1), 0, h->stream, h->ctl, st);
        CHK(sync_ctl(h));
        status = st;
        break;
      }
    }
    if (p.time_limit != 0.0) {
      double el = 0.0;
      CHK(collective_elapsed(h, t0, &el));
      if (el > p.time_limit) { CHK(enqueue_check(h, 0, 0)); CHK(sync_ctl(h)); status = COSMO_HIP_TIME_LIMIT_REACHED; break; }
    }
  }
  if (it + h->safeguarding_iter == p.max_iter && status != COSMO_HIP_TIME_LIMIT_REACHED) {   // solver.jl:173-176
    CHK(enqueue_check(h, 0, 0));
    CHK(sync_ctl(h));
    status = COSMO_HIP_MAX_ITER_REACHED;
  }
  *status_out = status;
  *iter_out = it;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_optimize(cosmo_hip_handle* h, cosmo_hip_result* res) {
  ENTER(h);
  if (!h->have_iterates || !res) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "optimize: set_iterates first");
  memset(res, 0, sizeof *res);
  const cosmo_hip_params& p = h->prm;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const auto t0 = std::chrono::steady_clock::now();
  int status = COSMO_HIP_UNDETERMINED;
  long long it = 0;
  long long acc_iter = -1;
  if (aa_enabled(h)) {
    CHK(optimize_accelerated(h, &status, &acc_iter, t0));
  } else {
    h->safeguarding_iter = 0;
    CHK(admm_init_enqueue(h));
  }
  while (!aa_enabled(h) && it < p.max_iter) {
    CHK(auto_rho_interval(h, it + 1, t0));                  // (the slices end at the termination checks: the rule is applied there)
    long long next = (it == 0) ? 1 : ((it / p.check_termination) + 1) * (long long)p.check_termination;
    next = std::min(next, next_inf_iter(h, it));
    if (next > p.max_iter) next = p.max_iter;
    if (p.time_limit != 0.0 && it > 0) {
      // the reference tests the time limit after EVERY iteration (solver.jl:351); the enqueue runs ahead in slices, so a slice is
      // cut to the number of iterations the measured pace fits into the remaining time (at least one)
      double el = 0.0;
      CHK(collective_elapsed(h, t0, &el));                // sharded: the same slice on every rank
      const double per_it = el / (double)it;
      const double left = p.time_limit - el;
      const long long fit = (left > 0.0 && per_it > 0.0) ? (long long)std::min(1e15, left / per_it) : 0;
      next = std::min(next, it + std::max<long long>(1, fit));
    }
    const bool check = (next % p.check_termination == 0) || next == 1;
    CHK(run_until(h, next, check ? 1 : -1));
    CHK(maybe_infeas_check(h, next));
    it = h->ctl_host->iter;
    if (h->ctl_host->status != 0) { status = h->ctl_host->status; break; }
    it = next;
    if (p.time_limit != 0.0) {
      double el = 0.0;
      CHK(collective_elapsed(h, t0, &el));
      if (el > p.time_limit) {  // solver.jl:351-354
        CHK(enqueue_check(h, 0, 0));
        CHK(sync_ctl(h));
        status = COSMO_HIP_TIME_LIMIT_REACHED;
        break;
      }
    }
  }
  if (!aa_enabled(h) && h->ctl_host->iter == p.max_iter && status != COSMO_HIP_TIME_LIMIT_REACHED) {
    // solver.jl:173-176: `if iter == max_iter` overrides ANY status decided in that very iteration (reference quirk kept)
    CHK(enqueue_check(h, 0, 0));
    CHK(sync_ctl(h));
    status = COSMO_HIP_MAX_ITER_REACHED;
  }
  CHK(launch_recover_mu(h));  // solver.jl:167
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const auto t1 = std::chrono::steady_clock::now();
  const Ctl* c = h->ctl_host;
  res->status = status;
  res->iter = (acc_iter >= 0) ? acc_iter + h->safeguarding_iter : c->iter;      // total_iter (src/solver.jl:196)
  res->safeguarding_iter = (acc_iter >= 0) ? h->safeguarding_iter : 0;
  res->kkt_iters_total = c->kkt_iters_total;
  res->kkt_solves = c->solves;
  res->cost = (status == COSMO_HIP_PRIMAL_INFEASIBLE) ? (double)INFINITY : (status == COSMO_HIP_DUAL_INFEASIBLE) ? -(double)INFINITY : (double)c->cost;   // solver.jl:339,345
  res->r_prim = (double)c->r_prim; res->r_dual = (double)c->r_dual; res->max_norm_prim = (double)c->max_norm_prim; res->max_norm_dual = (double)c->max_norm_dual;
  res->rho = (double)c->rho;
  res->n_rho_updates = c->n_rho_updates;
  for (int i = 0; i < COSMO_HIP_MAX_RHO_UPDATES && i < c->n_rho_updates; ++i) res->rho_updates[i] = (double)c->rho_updates[i];
  res->iter_time = std::chrono::duration<double>(t1 - t0).count();
  res->proj_time = h->kc_seconds[KC_Z] + h->kc_seconds[KC_SOC] + h->kc_seconds[KC_PSD];
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_iterates(cosmo_hip_handle* h, real* w, real* w_prev, real* s, real* mu) {
  ENTER(h);
  if (!h->have_iterates) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "get_iterates: set_iterates first");
  if (h->row_shard) return rs_get_iterates(h, w, w_prev, s, mu);      // every rank receives the GLOBAL vectors (all-gather of the row slices)
  const size_t N = (size_t)(h->n + h->m);
  if (mu) { CHK(launch_recover_mu(h)); CHK(d2h(h, mu, h->mu, (size_t)h->m)); }
  if (w) CHK(d2h(h, w, h->w, N));
  if (w_prev) CHK(d2h(h, w_prev, h->w_prev, N));
  if (s) CHK(d2h(h, s, h->s, (size_t)h->m));
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_cg_persist_stats(cosmo_hip_handle* h, int64_t out[8]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  out[0] = h->pcg_on ? 1 : 0; out[1] = h->pcg_W; out[2] = h->pcg_launches; out[3] = h->pcg_fallbacks;
  out[4] = out[5] = out[6] = 0; out[7] = h->pcg_cap;
  if (h->pcg_sync) {                                  // synchronisation words of the last launch: tickets, barrier arrivals, abort flag
    unsigned w[4] = {0, 0, 0, 0};
    if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
    HIPCHK(h, hipMemcpyAsync(w, h->pcg_sync, sizeof w, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out[4] = w[0]; out[5] = w[1]; out[6] = w[2];
    if (getenv("COSMO_HIP_PCG_TIMING")) {             // lab build (-DPCG_TIMING): phase clocks of workgroup 0 (100 MHz ticks) and iterations
      unsigned t[8];
      HIPCHK(h, hipMemcpy(t, h->pcg_sync + 8, sizeof t, hipMemcpyDeviceToHost));
      fprintf(stderr, "pcg timing (us per Krylov iteration over %u its): top %.2f  dirA %.2f  B2 %.2f  opapply %.2f  B3 %.2f  upd %.2f  B4 %.2f\n", t[7],
              t[0] / 100.0 / t[7], t[1] / 100.0 / t[7], t[2] / 100.0 / t[7], t[3] / 100.0 / t[7], t[4] / 100.0 / t[7], t[5] / 100.0 / t[7], t[6] / 100.0 / t[7]);
    }
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_kkt_solution(cosmo_hip_handle* h, real* sol) {
  ENTER(h);
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "get_kkt_solution: not available on a row-sharded handle");
  if (!h->have_params || !sol) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "get_kkt_solution: not set up");
  CHK(d2h(h, sol, h->x_tl, (size_t)h->n));
  CHK(d2h(h, sol + h->n, h->nu, (size_t)h->m));
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_set_setup_time(cosmo_hip_handle* h, double seconds) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (!(seconds >= 0.0)) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_setup_time: need seconds >= 0");
  h->prm.setup_time = seconds;
  return COSMO_HIP_OK;
}
extern "C" int32_t cosmo_hip_get_rho_interval(cosmo_hip_handle* h, int64_t out[2]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  out[0] = h->prm.adaptive_rho_interval; out[1] = h->auto_rho_fixed_at;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_stats(cosmo_hip_handle* h, int64_t out[8]) {
  ENTER(h);
  if (!out) return COSMO_HIP_ERR_INVALID;
  CHK(sync_ctl(h));
  const Ctl* c = h->ctl_host;
  out[0] = c->iter; out[1] = c->solves; out[2] = c->kkt_iters_total; out[3] = h->stalls;
  out[4] = h->spmv_calls[0]; out[5] = h->spmv_calls[1]; out[6] = h->spmv_calls[2]; out[7] = c->n_rho_updates;
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------------------------------------------------
int32_t time_op_apply(cosmo_hip_handle* h, int reps, double* avg_seconds);  // kernels.hip

extern "C" int32_t cosmo_hip_time_spmv(cosmo_hip_handle* h, int32_t which, int32_t reps, double* avg_seconds,
                                       double* algorithmic_bytes) {
  ENTER(h);
  if (!h->have_problem || reps <= 0 || !avg_seconds) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "time_spmv: bad arguments");
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "time_spmv: not available on a row-sharded handle (A / A' are row slices, [P | A'] is dropped)");
  const long long n = h->n, m = h->m;
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  double bytes = 0.0;
  // 12 B per nonzero + 4 B per row pointer + 8 B per input and output vector element
  if (which == COSMO_HIP_MAT_A) bytes = 12.0 * h->A.nnz + 4.0 * (m + 1) + 8.0 * n + 8.0 * m;
  else if (which == COSMO_HIP_MAT_AT) bytes = 12.0 * h->AT.nnz + 4.0 * (n + 1) + 8.0 * m + 8.0 * n;
  else if (which == COSMO_HIP_MAT_P) bytes = 12.0 * h->P.nnz + 4.0 * (n + 1) + 16.0 * n;
  else if (which == 3) bytes = 12.0 * h->PT.nnz + 8.0 * (n + 1) + 8.0 * (n + m) + 8.0 * n;  // + split pointers
  else return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "time_spmv: bad matrix id");
  // warm up + timed back-to-back launches on the handle's stream
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) HIPCHK(h, hipEventRecord(e0, h->stream));
    const int R = pass == 0 ? 3 : reps;
    for (int i = 0; i < R; ++i) {
      if (which == COSMO_HIP_MAT_A) CHK(launch_spmv_plain(h, h->A, h->u, h->tmp_m));
      else if (which == COSMO_HIP_MAT_AT) CHK(launch_spmv_plain(h, h->AT, h->tmp_m, h->c));
      else if (which == COSMO_HIP_MAT_P) CHK(launch_spmv_plain(h, h->P, h->u, h->c));
      else CHK(time_op_apply(h, 1, nullptr));
    }
    if (pass == 1) HIPCHK(h, hipEventRecord(e1, h->stream));
  }
  HIPCHK(h, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  *avg_seconds = (double)ms * 1e-3 / reps;
  if (algorithmic_bytes) *algorithmic_bytes = bytes;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_time_krylov(cosmo_hip_handle* h, int32_t reps, double* avg_seconds, double* algorithmic_bytes, int32_t* launches_per_iteration) {
  ENTER(h);
  if (!h->have_params || !h->have_iterates || reps <= 0 || !avg_seconds) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "time_krylov: set up the loop first");
  if (h->prm.kkt_kind != COSMO_HIP_KKT_CG || (h->cg_sr && !h->op_fold) || h->pcg_on || h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "time_krylov: the CG recurrences of the loop (literal, Jacobi, one-launch single-reduction on the assembled operator) on an unsharded handle only");
  CHK(sync_ctl(h));
  if (h->ctl_host->halt) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "time_krylov: the loop is halted");
  const long long n = h->n, m = h->m;
  real* keep = nullptr;
  HIPCHK(h, hipMalloc((void**)&keep, sizeof(real) * (size_t)std::max<long long>(n, 1)));
  HIPCHK(h, hipMemcpyAsync(keep, h->x_tl, sizeof(real) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  const long long sp[3] = {h->spmv_calls[0], h->spmv_calls[1], h->spmv_calls[2]};
  int32_t rc = enqueue_y2_only(h);                             // resets the per-solve flags (y2 = rho .* ls_s is recomputed to the same values)
  if (rc == COSMO_HIP_OK) rc = enqueue_cg_start(h, 1, R(0.0)); // tolerance 0: every one of the `reps` iterations does full work
  if (rc == COSMO_HIP_OK && h->cg_sr) rc = sr_enqueue_start(h, 1);
  if (rc == COSMO_HIP_OK && hipEventRecord(e0, h->stream) != hipSuccess) rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipEventRecord failed");
  const int likely = h->cg_k_likely;
  h->cg_k_likely = 0x7fffffff;
  if (rc == COSMO_HIP_OK) rc = h->cg_sr ? sr_enqueue_iterations(h, 1, 0, reps) : enqueue_cg_iterations(h, 1, 0, reps);
  h->cg_k_likely = likely;
  if (rc == COSMO_HIP_OK && hipEventRecord(e1, h->stream) != hipSuccess) rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipEventRecord failed");
  if (rc == COSMO_HIP_OK && hipEventSynchronize(e1) != hipSuccess) rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipEventSynchronize failed");
  float ms = 0.f;
  if (rc == COSMO_HIP_OK) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  // restore: the warm start of the next solve, the multiplication counters; r / u / c / the records are rewritten by every solve start
  (void)hipMemcpyAsync(h->x_tl, keep, sizeof(real) * (size_t)n, hipMemcpyDeviceToDevice, h->stream);
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(keep);
  h->spmv_calls[0] = sp[0]; h->spmv_calls[1] = sp[1]; h->spmv_calls[2] = sp[2];
  if (rc != COSMO_HIP_OK) return rc;
  *avg_seconds = (double)ms * 1e-3 / reps;
  double bytes; int nl;
  if (h->op_fold) {
    const FoldPlan* f = (const FoldPlan*)h->fold;
    bytes = 12.0 * (double)f->nnz_full + 4.0 * (n + 1) + 16.0 * n + 8.0 * 10.0 * n;    // B_spmv(M) of the FULLY assembled operator (the unit since round 2; a partially
                                                                                        // assembled operator streams fewer entries for the same product) + B_cgvec (n-side); Jacobi adds dinv: + 8 n
    if (h->cg_jacobi) bytes += 8.0 * n;
    nl = 2;
    if (h->cg_sr) nl = 1;      // one-launch recurrence: the SAME algorithmic bytes (SURVEY 8d prices a Krylov iteration of the reference: operator + 10 n-vectors; the
                               // records {r, w, s, p} read and written once by their owners + x read and written are 10 n words too)
  } else {
    const CsrDev& Ao = h->op_split ? h->Am : h->A;
    const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
    bytes = (12.0 * Ao.nnz + 4.0 * (Ao.nrows + 1) + 8.0 * n + 8.0 * Ao.nrows) + (12.0 * PTo.nnz + 8.0 * (n + 1) + 8.0 * (n + Ao.nrows) + 8.0 * n) + 8.0 * (10.0 * n + m);
    nl = h->cg_ru ? 3 : 4;
  }
  if (algorithmic_bytes) *algorithmic_bytes = bytes;
  if (launches_per_iteration) *launches_per_iteration = nl;
  return COSMO_HIP_OK;
}

extern "C" const char* cosmo_hip_kkt_recurrence(cosmo_hip_handle* h) {
  if (!h || !h->have_params) return "not set up";
  static const char* sr_names[] = {"", "k_sr_M<1>", "k_sr_M<2>", "k_sr_M<3>", "k_sr_M<4>", "", "", "", "k_sr_M<8>"};
  static const char* pair_names[] = {"", "k_cg_dirM<1, false> + k_cg_upd<false>", "k_cg_dirM<2, false> + k_cg_upd<false>", "k_cg_dirM<3, false> + k_cg_upd<false>",
                                     "k_cg_dirM<4, false> + k_cg_upd<false>", "", "", "", "k_cg_dirM<8, false> + k_cg_upd<false>"};
  static const char* updf_names[] = {"", "k_cg_dirM<1, false> + k_cg_updF", "k_cg_dirM<2, false> + k_cg_updF", "k_cg_dirM<3, false> + k_cg_updF", "k_cg_dirM<4, false> + k_cg_updF", "", "", "",
                                     "k_cg_dirM<8, false> + k_cg_updF"};
  static const char* pc_names[] = {"", "k_cg_dirM<1, true> + k_cg_upd<true>", "k_cg_dirM<2, true> + k_cg_upd<true>", "k_cg_dirM<3, true> + k_cg_upd<true>",
                                   "k_cg_dirM<4, true> + k_cg_upd<true>", "", "", "", "k_cg_dirM<8, true> + k_cg_upd<true>"};
  static thread_local char buf[256];
  if (h->prm.kkt_kind == COSMO_HIP_KKT_MINRES) return "minres on the full KKT system (csrc/minres.hip)";
  if (h->prm.kkt_kind == COSMO_HIP_KKT_MINRES_REDUCED) return "minres on the reduced system (csrc/minres.hip)";
  const FoldPlan* f = (const FoldPlan*)h->fold;
  const int sl = (h->op_fold && f) ? ((f->slots >= 1 && f->slots <= 4) ? f->slots : 8) : 0;
  if (h->pcg_on) return "cg: literal recurrence in one persistent launch (csrc/cg_persist.hip, opt-in)";
  if (h->cg_jacobi) { snprintf(buf, sizeof buf, "cg: Jacobi-preconditioned recurrence on the assembled operator (opt-in), %s", pc_names[sl]); return buf; }
  if (h->cg_sr && h->op_fold) {
    snprintf(buf, sizeof buf, "cg: one-launch single-reduction recurrence on the assembled operator%s, %s", h->cg_sr_auto ? " (lab switch COSMO_HIP_CG_SR_DEFAULT=1)" : " (opt-in kkt_kind CG_SR)", sr_names[sl]);
    return buf;
  }
  if (h->cg_sr) return "cg: single-reduction recurrence, two launches per iteration (kkt_kind CG_SR), k_sr_update_A + k_sr_op";
  if (h->op_fold && f->nd > 0) { snprintf(buf, sizeof buf, "cg: literal recurrence on the partially assembled operator (%d rows of A kept factored), two launches per iteration, %s", f->nd, updf_names[sl]); return buf; }
  if (h->op_fold) { snprintf(buf, sizeof buf, "cg: literal recurrence on the assembled operator, two launches per iteration, %s", pair_names[sl]); return buf; }
  return h->cg_ru ? "cg: literal recurrence, three launches per iteration, k_cg_dirA + k_op_apply + k_cg_upd<false>"
                  : "cg: literal recurrence, four launches per iteration, k_cg_dir + k_spmv_A_rho + k_op_apply + k_cg_upd<false>";
}

extern "C" int32_t cosmo_hip_set_profiling(cosmo_hip_handle* h, int32_t on) {
  ENTER(h);
  CHK(prof_collect(h));
  h->profiling = on == 1;
  h->exact_launches = on != 0;
  if (on) { memset(h->kc_seconds, 0, sizeof h->kc_seconds); memset(h->kc_launches, 0, sizeof h->kc_launches); }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_kernel_times(cosmo_hip_handle* h, double seconds[COSMO_HIP_NUM_KERNEL_CLASSES],
                                              int64_t launches[COSMO_HIP_NUM_KERNEL_CLASSES]) {
  ENTER(h);
  CHK(prof_collect(h));
  for (int i = 0; i < COSMO_HIP_NUM_KERNEL_CLASSES; ++i) {
    if (seconds) seconds[i] = h->kc_seconds[i];
    if (launches) launches[i] = h->kc_launches[i];
  }
  return COSMO_HIP_OK;
}

extern "C" const char* cosmo_hip_kernel_class_name(int32_t k) {
  static const char* names[COSMO_HIP_NUM_KERNEL_CLASSES] = {
      "admm_z(copy+simple cones)", "proj_soc", "proj_psd", "admm_x_rhs", "spmv_AT(cg rhs)", "spmv_A(rho.*A v)",
      "op_apply([P|A'] fused)", "cg_direction", "cg_update", "tail(A x_tl, s_tl, w)", "check_primal(A)",
      "check_dual([P|A'])", "check_final", "rho_apply", "minres_vec", "other"};
  if (k < 0 || k >= COSMO_HIP_NUM_KERNEL_CLASSES) return "?";
  return names[k];
}
