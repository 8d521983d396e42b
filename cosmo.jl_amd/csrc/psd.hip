// psd.hip -- projection onto the positive semidefinite cone (PsdConeTriangle / PsdCone), gfx950.
//
// Reference semantics (src/convexset.jl:219-263, 303-321, 402-412): X+ = sum_{lambda_j > 0} lambda_j z_j z_j' from the
// symmetric eigendecomposition of X (upper triangle), rank nnz_lambda = #{lambda_j > 0}; svec layout column-major upper
// triangle with off-diagonals scaled by sqrt(2) (Appendix C of SURVEY.md).
//
// MI355X design.  The eigendecomposition is a Jacobi method, chosen because it is all level-3 / embarrassingly
// parallel work for a 64-wide wave machine and needs no sequential tridiagonal QR chain:
//   d <= 16      one WAVE per cone: cyclic two-sided Jacobi on the 16x16 (zero padded) matrix held in LDS, 8 disjoint
//                rotations per round, all 64 lanes apply them.
//   16 < d<=256  one WORKGROUP per cone: block one-sided (Hestenes) Jacobi on G = X + c I  (c = ||X||_F >= rho(X), so G is
//                PSD and its SVD is its eigendecomposition: no eigenvalue-sign ambiguity).  Columns are grouped in blocks
//                of 8; per step every wave owns one block pair (16 columns): Gram matrix W = P'P with fp64 MFMA
//                (v_mfma_f64_16x16x4_f64), one sweep of the 16x16 Jacobi on W accumulating J, panel update P <- P J with
//                MFMA.  Round-robin ordering, workgroup barrier between steps, sweeps until no rotation fires.
//   d > 256      the same step as a multi-workgroup kernel (one workgroup per block pair, rows split over its waves),
//                one launch per step, host checks the convergence flag once per sweep.
// Then sigma_k = ||g_k||, lambda_k = sigma_k - c, and X+ = Ghat Ghat' with ghat_k = g_k sqrt(lambda_k)/sigma_k for
// lambda_k > 0 -- a SYRK on MFMA fused with the svec write-out (the reference's rank_k_update!, :243-263).
#include "device_utils.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>

#include "psd_internal.h"
#include "psd16.h"
#include "psdwg.h"

// ---------------------------------------------------------------------------------------------------------------------
// d <= 16: one wave per cone, two-sided Jacobi directly on X (psd16.h: the same routine the batch kernels call)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_psd_tiny(const Ctl* __restrict__ ctl, int guard, int ncones,
                                                       const int* __restrict__ list, const PsdConeDev* __restrict__ cones,
                                                       real* __restrict__ s, int* __restrict__ rank, int* __restrict__ flags,
                                                       int mode, real sign, real* __restrict__ eigmin) {
  // mode 0: project in place.  mode 1: only the smallest eigenvalue of sign * mat(x) (definiteness tests of infeasibility.jl)
  if (guard && ctl->halt) return;
  __shared__ __attribute__((aligned(16))) unsigned char wsb[COSMO_BS / 64][(PSD16_WS_BYTES + 15) / 16 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int widx = blockIdx.x * (COSMO_BS / 64) + wv;
  if (widx >= ncones) return;
  const PsdConeDev cn = cones[list[widx]];
  const Psd16Ws ws = psd16_ws_at(wsb[wv]);
  int nonconv = 0;
  real lm = 0.0;
  const int rk = psd16_wave(s + cn.off, cn.d, cn.kind, ws, lane, mode, sign, &lm, &nonconv);
  if (nonconv && lane == 0) atomicOr(&flags[1], 1);            // did not converge
  if (mode == 1) { if (lane == 0) eigmin[list[widx]] = lm; return; }
  if (lane == 0) rank[list[widx]] = rk;
}

// ---------------------------------------------------------------------------------------------------------------------
// d > 16: build G = X + c I (full symmetric, zero padded), c = ||X||_F  (one workgroup per cone, any block size)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_psd_populate(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ list,
                                                           const PsdConeDev* __restrict__ cones, const real* __restrict__ s,
                                                           real* __restrict__ G, real* __restrict__ cshift, real sign, int upper_only) {
  // upper_only: the definiteness tests read Hermitian(X, 'U') of the square layout WITHOUT symmetrising (is_pos_def!,
  // src/algebra.jl:226-233) -- delta_y of a PsdCone is not symmetric in general -- while project! symmetrises first (:201-208)
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const int ci = list[blockIdx.y];
  const PsdConeDev cn = cones[ci];
  const real* x = s + cn.off;
  const int d = cn.d;
  real* g = G + cn.goff;
  const real isq2 = 1.0 / sqrt(2.0);
  // ||X||_F: for the svec layout it is ||x||_2 (the scaling makes svec an isometry)
  real acc = 0.0;
  if (cn.kind == COSMO_HIP_PSD_TRIANGLE) {
    const long long len = (long long)d * (d + 1) / 2;
    for (long long k = threadIdx.x; k < len; k += COSMO_BS) { const real v = x[k]; acc += v * v; }
  } else {
    for (long long k = threadIdx.x; k < (long long)d * d; k += COSMO_BS) {
      const int i = (int)(k % d), j = (int)(k / d);
      const int a = i < j ? i : j, b = i < j ? j : i;
      const real v = upper_only ? x[(long long)b * d + a] : (x[(long long)j * d + i] + x[(long long)i * d + j]) / R(2.0);
      acc += v * v;
    }
  }
  const real c = sqrt(block_sum(acc, red));
  if (threadIdx.x == 0 && blockIdx.x == 0) cshift[ci] = c;
  // columns are distributed over blockIdx.x
  for (int j = blockIdx.x; j < cn.ncp; j += gridDim.x) {
    for (int i = threadIdx.x; i < cn.ld; i += COSMO_BS) {
      real v = 0.0;
      if (i < d && j < d) {
        const int a = i < j ? i : j, b = i < j ? j : i;
        if (cn.kind == COSMO_HIP_PSD_TRIANGLE) {
          const real t = x[svec_idx(a, b)];
          v = (a == b) ? t : isq2 * t;
        } else {
          v = upper_only ? x[(long long)b * d + a] : (x[(long long)b * d + a] + x[(long long)a * d + b]) / R(2.0);
        }
        v = v * sign;
        if (i == j) v += c;
      }
      g[(long long)j * cn.ld + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16 < d <= 256: whole Jacobi process of one cone in one workgroup (nb/2 <= 16 waves)
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_psd_jacobi_wg(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ list,
                                                           const PsdConeDev* __restrict__ cones, real* __restrict__ G,
                                                           const real* __restrict__ cshift, int* __restrict__ flags, real tolf, int dbg) {
  if (guard && ctl->halt) return;
  __shared__ __attribute__((aligned(16))) unsigned char wsb[NW * PSDWG_WS_STRIDE];
  __shared__ int any_rot;
  const int ci = list[blockIdx.x];
  const PsdConeDev cn = cones[ci];
  const int sweep = psdwg_jacobi<NW>(G + cn.goff, cn.ld, cn.nb, cn.d, cshift[ci], tolf, dbg, wsb, &any_rot);      // psdwg.h (shared with the batch kernels)
  if (threadIdx.x == 0) {
    atomicMax(&flags[2], sweep + 1);
    if (sweep >= PSD_MAX_SWEEPS) atomicOr(&flags[1], 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// d > 256: one step of the same process, one workgroup per block pair, panel rows split over the waves
// grid = (npairs_max, nlarge);  flags[0] accumulates "some rotation fired in this sweep"
// ---------------------------------------------------------------------------------------------------------------------
#define PSD_STEP_WAVES 8
__global__ __launch_bounds__(PSD_STEP_WAVES * 64) void k_psd_step(const int* __restrict__ list, const PsdConeDev* __restrict__ cones,
                                                                  real* __restrict__ G, const real* __restrict__ cshift,
                                                                  int st, int* __restrict__ flags, real tolf) {
  __shared__ real Wp[PSD_STEP_WAVES][16 * WLD];
  __shared__ real Ws[16 * WLD];
  __shared__ real Js[16 * WLD];
  __shared__ real cas[16], cbs[16];
  __shared__ int parts[16];
  __shared__ int cols[16];
  __shared__ int rot_s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ci = list[blockIdx.y];
  const PsdConeDev cn = cones[ci];
  const int nb = cn.nb, npairs = nb / 2;
  if ((int)blockIdx.x >= npairs || st >= nb - 1) return;
  const int full = (st < 0) ? 1 : 0;
  real* g = G + cn.goff;
  const real c = cshift[ci];
  const real tol = tolf * (real)cn.d * PSD_EPS;
  const real tiny = (tol * c) * (tol * c);
  int I, Jb;
  if (full) { I = 2 * blockIdx.x; Jb = I + 1; }
  else rr_pair(nb, st, blockIdx.x, I, Jb);
  if (threadIdx.x < 16) cols[threadIdx.x] = (threadIdx.x < 8) ? (I * 8 + threadIdx.x) : (Jb * 8 + threadIdx.x - 8);
  __syncthreads();
  // rows split in 16-row chunks over the waves
  const int nch = cn.ld / 16;
  const int per = (nch + PSD_STEP_WAVES - 1) / PSD_STEP_WAVES;
  const int r0 = min(nch, wv * per) * 16, r1 = min(nch, (wv + 1) * per) * 16;
  const v4d w = panel_gram(g, cn.ld, cols[lane & 15], r0, r1, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) Wp[wv][ACC_ROW(lane, r) * WLD + (lane & 15)] = w[r];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ACC_ROW(lane, r), j = lane & 15;
      real a = 0.0;
      for (int q = 0; q < PSD_STEP_WAVES; ++q) a += Wp[q][i * WLD + j];
      Ws[i * WLD + j] = a;
      Js[i * WLD + j] = (i == j) ? 1.0 : 0.0;
    }
    wave_lds_fence();
    int rot = 0;
    if (gram_needs_work(Ws, tol, tiny, !full, lane)) rot = jacobi16_sweep(Ws, Js, parts, cas, cbs, tol, tiny, 0, full ? 15 : 8, lane);
    if (lane == 0) { rot_s = rot; if (rot) atomicOr(&flags[0], 1); }
  }
  __syncthreads();
  if (rot_s) {
    real jt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) jt[t] = Js[((lane >> 4) + 4 * t) * WLD + (lane & 15)];
    panel_update(g, cn.ld, cols, jt, r0, r1, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// sigma_k = ||g_k|| ; lambda_k = sigma_k - c ; scale_k = sqrt(lambda_k) / sigma_k for lambda_k > 0 else 0 ; rank
// grid = (1, ncones in list), one workgroup per cone; columns over waves
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_psd_colscale(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ list,
                                                           const PsdConeDev* __restrict__ cones, real* __restrict__ G,
                                                           const real* __restrict__ cshift, real* __restrict__ colw,
                                                           int* __restrict__ rank) {
  if (guard && ctl->halt) return;
  const int ci = list[blockIdx.y];
  const PsdConeDev cn = cones[ci];
  real* g = G + cn.goff;
  const real c = cshift[ci];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int mycnt = 0;
  for (int j = blockIdx.x * (COSMO_BS / 64) + wv; j < cn.ncp; j += gridDim.x * (COSMO_BS / 64)) {
    real* col = g + (long long)j * cn.ld;
    real a = 0.0;
    for (int i = lane; i < cn.ld; i += 64) { const real v = col[i]; a += v * v; }
    const real sig = sqrt(wave_sum(a));
    const real lam = sig - c;
    real f = 0.0;
    if (j < cn.d && lam > R(0.0) && sig > R(0.0)) { f = sqrt(lam) / sig; mycnt += 1; }
    // scale the column in place: ghat_k = g_k sqrt(lambda_k) / sigma_k  (rank_k_update!, convexset.jl:248-256)
    for (int i = lane; i < cn.ld; i += 64) col[i] = col[i] * f;
    if (lane == 0) colw[cn.coff + j] = lam;
  }
  if (lane == 0 && mycnt) atomicAdd(&rank[ci], mycnt);    // integer count: order independent
}

// smallest eigenvalue of every cone of the list: min_k (||g_k|| - c) over the real columns (one workgroup per cone)
__global__ __launch_bounds__(COSMO_BS) void k_psd_eigmin(const int* __restrict__ list, const PsdConeDev* __restrict__ cones,
                                                         const real* __restrict__ G, const real* __restrict__ cshift,
                                                         real* __restrict__ eigmin) {
  __shared__ real red[COSMO_BS / 64];
  const int ci = list[blockIdx.x];
  const PsdConeDev cn = cones[ci];
  const real* g = G + cn.goff;
  const real c = cshift[ci];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  real lm = INFINITY;
  for (int j = wv; j < cn.d; j += COSMO_BS / 64) {
    const real* col = g + (long long)j * cn.ld;
    real a = 0.0;
    for (int i = lane; i < cn.ld; i += 64) { const real v = col[i]; a += v * v; }
    lm = fmin(lm, sqrt(wave_sum(a)) - c);
  }
  lm = -block_max(-lm, red);
  if (threadIdx.x == 0) eigmin[ci] = lm;
}

// X+ = Ghat Ghat' (upper tiles) on MFMA, written straight into s.  grid = (ntiles_max, ncones); one wave per 16x16 tile.
__global__ __launch_bounds__(COSMO_BS) void k_psd_syrk(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ list,
                                                       const PsdConeDev* __restrict__ cones, const real* __restrict__ G,
                                                       real* __restrict__ s) {
  if (guard && ctl->halt) return;
  const int ci = list[blockIdx.y];
  const PsdConeDev cn = cones[ci];
  const int nt = cn.ld / 16;                         // tiles per side
  const int ntiles = nt * (nt + 1) / 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const real* g = G + cn.goff;
  real* x = s + cn.off;
  const int d = cn.d;
  const real sq2 = sqrt(2.0);
  for (int t = blockIdx.x * (COSMO_BS / 64) + wv; t < ntiles; t += gridDim.x * (COSMO_BS / 64)) {
    // unrank t -> (ti <= tj), column-major over upper tiles
    int tj = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) / 2.0);
    while ((long long)tj * (tj + 1) / 2 > t) --tj;
    while ((long long)(tj + 1) * (tj + 2) / 2 <= t) ++tj;
    const int ti = t - tj * (tj + 1) / 2;
    // D[a][b] = X[16 ti + b][16 tj + a] = sum_k Ghat[16 tj + a][k] Ghat[16 ti + b][k]
    const real* pa = g + 16 * tj + (lane & 15) + (long long)(lane >> 4) * cn.ld;   // A[a = l&15][k = l>>4]
    const real* pb = g + 16 * ti + (lane & 15) + (long long)(lane >> 4) * cn.ld;   // B[k = l>>4][b = l&15]
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < cn.ncp; k += 4) {
      const real a = pa[(long long)k * cn.ld];
      const real b = pb[(long long)k * cn.ld];
      acc = MFMA_REAL(a, b, acc);
    }
    const int i = 16 * ti + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * tj + ACC_ROW(lane, r);
      if (i < d && j < d) {
        if (cn.kind == COSMO_HIP_PSD_TRIANGLE) {
          if (i <= j) x[svec_idx(i, j)] = (i == j) ? acc[r] : sq2 * acc[r];
        } else {
          if (i <= j) { x[(long long)j * d + i] = acc[r]; x[(long long)i * d + j] = acc[r]; }   // mirror (convexset.jl:316-318)
        }
      }
    }
  }
}

// =====================================================================================================================
// host side
// =====================================================================================================================
template <class T>
static int32_t up(cosmo_hip_handle* h, T** d, const std::vector<T>& v) {
  if (*d) { (void)hipFree(*d); *d = nullptr; }
  HIPCHK(h, hipMalloc((void**)d, std::max<size_t>(1, v.size()) * sizeof(T)));
  if (!v.empty()) HIPCHK(h, hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return COSMO_HIP_OK;
}

void psd_plan_destroy(cosmo_hip_handle* h) {
  PsdPlan* p = h->psd;
  if (!p) return;
  if (p->d_cones) (void)hipFree(p->d_cones);
  if (p->d_tiny) (void)hipFree(p->d_tiny);
  if (p->d_large) (void)hipFree(p->d_large);
  for (auto q : p->d_wg_groups) if (q) (void)hipFree(q);
  for (auto q : p->d_pj_groups) if (q) (void)hipFree(q);
  if (p->G) (void)hipFree(p->G);
  if (p->colw) (void)hipFree(p->colw);
  if (p->cshift) (void)hipFree(p->cshift);
  if (p->rank) (void)hipFree(p->rank);
  if (p->flags) (void)hipFree(p->flags);
  if (p->eigmin) (void)hipFree(p->eigmin);
  delete p;
  h->psd = nullptr;
  polar_plan_destroy(h);
}

int32_t psd_plan_create(cosmo_hip_handle* h) {
  psd_plan_destroy(h);
  PsdPlan* p = new PsdPlan();
  h->psd = p;
  const ConeTable& C = h->cones;
  long long goff = 0;
  int coff = 0;
  for (size_t k = 0; k < C.type.size(); ++k) {
    if (C.type[k] != COSMO_HIP_PSD_SQUARE && C.type[k] != COSMO_HIP_PSD_TRIANGLE && C.type[k] != COSMO_HIP_PSD_TRIANGLE_COMPLEX) continue;
    if (C.dim[k] <= 1) continue;
    if (!cone_owned(h, (long long)k)) continue;   // clique sharding: another rank projects this cone
    if (C.type[k] == COSMO_HIP_PSD_TRIANGLE_COMPLEX) {
      // Hermitian r x r cone: projected through its real symmetric embedding [[A, -B], [B, A]] of side 2r by the matrix-sign
      // paths (psd_polar.hip); never enters the Jacobi size classes
      PsdConeDev cc;
      cc.kind = C.type[k]; cc.off = (int)C.off[k];
      cc.d = 2 * (int)llround(sqrt((real)C.dim[k]));
      cc.ld = ((cc.d + 15) / 16) * 16; cc.nb = 0; cc.ncp = 0; cc.goff = 0; cc.coff = 0; cc.cone_index = (int)k;
      p->cplx.push_back((int)p->cones.size());
      p->cones.push_back(cc);
      continue;
    }
    PsdConeDev cn;
    cn.kind = C.type[k];
    cn.off = (int)C.off[k];
    if (cn.kind == COSMO_HIP_PSD_SQUARE) cn.d = (int)llround(sqrt((real)C.dim[k]));
    else cn.d = (int)((llround(floor(sqrt(1.0 + 8.0 * (double)C.dim[k]))) - 1) / 2);
    while ((long long)cn.d * (cn.d + 1) / 2 > C.dim[k] && cn.kind == COSMO_HIP_PSD_TRIANGLE) --cn.d;
    if (cn.kind == COSMO_HIP_PSD_TRIANGLE && (long long)cn.d * (cn.d + 1) / 2 != C.dim[k])
      return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "PsdConeTriangle dimension %lld is not triangular", (long long)C.dim[k]);
    cn.ld = ((cn.d + 15) / 16) * 16;
    cn.nb = (cn.d + 7) / 8;
    if (cn.nb & 1) cn.nb += 1;
    cn.ncp = cn.nb * 8;
    if (cn.ncp < cn.ld) { cn.ncp = cn.ld; cn.nb = cn.ncp / 8; }   // SYRK walks k over ncp and tiles over ld
    cn.goff = goff;
    cn.coff = coff;
    cn.cone_index = (int)k;
    const int idx = (int)p->cones.size();
    if (cn.d <= 16) p->tiny.push_back(idx);
    else {
      goff += (long long)cn.ld * cn.ncp;
      coff += cn.ncp;
      if (cn.d <= 256) p->wg.push_back(idx); else p->large.push_back(idx);
    }
    p->cones.push_back(cn);
  }
  if (const char* e = getenv("COSMO_HIP_PSD_TOL_FACTOR")) p->tol_factor = atof(e);
  if (const char* e = getenv("COSMO_HIP_PSD_DEBUG")) p->dbg = atoi(e);   // timing ablations only (results are wrong)
  if (p->cones.empty()) return COSMO_HIP_OK;
  p->gsize = goff; p->ncolw = coff;
  CHK(up(h, &p->d_cones, p->cones));
  CHK(up(h, &p->d_tiny, p->tiny));
  CHK(up(h, &p->d_large, p->large));
  // workgroup class: group by waves needed (nb/2 rounded up to 2, 4, 8, 16), largest first for load balance
  // two launch classes (16 or 4 waves per workgroup) so that the big cones of a mixed batch run concurrently
  const int classes[2] = {16, 4};
  for (int c = 0; c < 2; ++c) {
    std::vector<int> grp;
    for (int idx : p->wg) {
      const int need = p->cones[idx].nb / 2;
      const int lo = (c == 1) ? 0 : classes[c + 1];
      if (need <= classes[c] && need > lo) grp.push_back(idx);
    }
    if (grp.empty()) continue;
    std::sort(grp.begin(), grp.end(), [&](int a, int b) { return p->cones[a].d > p->cones[b].d; });
    p->wg_waves.push_back(classes[c]);
    p->wg_groups.push_back(grp);
    int* dptr = nullptr;
    CHK(up(h, &dptr, grp));
    p->d_wg_groups.push_back(dptr);
  }
  // projection-time split of the wg class: cones with d > polar_min go to the batched matrix-sign path (psd_polar.hip), whose
  // ~80 launches cost the same for any number of cones, while a d = 200 Jacobi workgroup needs ~300 dependent tournament steps.
  // Round 2: 64 -> 16.  A 16 < d <= 64 cone is ONE extra 64 x 64 tile in launches that already exist, whereas the one-workgroup
  // Jacobi kernel of those cones was a 0.4-0.55 ms serial phase of BASELINE config 5 (100 of its 400 cliques): 123.9 -> 132.3 it/s
  // (profiles/r02_cfg5_polar_batch_min_and_fold.json); and the sign path is the more accurate one (5e-15 vs 1e-11 ||X||_F).
  // COSMO_HIP_POLAR_BATCH_MIN=256 sends every workgroup-class cone back to the Jacobi kernels (kept under test that way).
  {
    int polar_min = (h->psd_mode == 1) ? 256 : 16;        // cosmo_hip_set_psd_projection(EIGEN): every workgroup-class cone on the Jacobi kernels
    if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_MIN")) polar_min = atoi(e);
    p->polar_batch.clear();
    for (int idx : p->wg) if (p->cones[idx].d > polar_min) p->polar_batch.push_back(idx);
    for (size_t gi = 0; gi < p->wg_groups.size(); ++gi) {
      std::vector<int> grp;
      for (int idx : p->wg_groups[gi]) if (p->cones[idx].d <= polar_min) grp.push_back(idx);
      if (grp.empty()) continue;
      p->pj_waves.push_back(p->wg_waves[gi]);
      p->pj_groups.push_back(grp);
      int* dptr = nullptr;
      CHK(up(h, &dptr, grp));
      p->d_pj_groups.push_back(dptr);
    }
  }
  HIPCHK(h, hipMalloc((void**)&p->G, std::max<long long>(1, p->gsize) * sizeof(real)));
  HIPCHK(h, hipMalloc((void**)&p->colw, std::max(1, p->ncolw) * sizeof(real)));
  HIPCHK(h, hipMalloc((void**)&p->cshift, p->cones.size() * sizeof(real)));
  HIPCHK(h, hipMalloc((void**)&p->rank, p->cones.size() * sizeof(int)));
  HIPCHK(h, hipMalloc((void**)&p->flags, 4 * sizeof(int)));
  HIPCHK(h, hipMalloc((void**)&p->eigmin, p->cones.size() * sizeof(real)));
  HIPCHK(h, hipMemset(p->rank, 0, p->cones.size() * sizeof(int)));
  HIPCHK(h, hipMemset(p->flags, 0, 4 * sizeof(int)));
  return polar_plan_create(h);   // d > 256: matrix-sign iteration on the matrix cores (psd_polar.hip)
}

// host-paced Jacobi sweeps of the multi-workgroup path (one launch per tournament step, flag read once per sweep)
static int32_t psd_large_sweeps(cosmo_hip_handle* h, int n, int nbmax) {
  PsdPlan* p = h->psd;
  int sweep = 0;
  for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
    HIPCHK(h, hipMemsetAsync(p->flags, 0, sizeof(int), h->stream));
    for (int st = -1; st < nbmax - 1; ++st)
      hipLaunchKernelGGL(k_psd_step, dim3(nbmax / 2, n), dim3(PSD_STEP_WAVES * 64), 0, h->stream, p->d_large, p->d_cones, p->G,
                         p->cshift, st, p->flags, p->tol_factor);
    int fl = 0;
    HIPCHK(h, hipMemcpyAsync(&fl, p->flags, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!fl) break;
  }
  p->last_large_sweeps = sweep + 1;
  if (sweep >= PSD_MAX_SWEEPS) return cosmo_fail(h, COSMO_HIP_ERR_EIG, "Jacobi eigensolver did not converge in %d sweeps", PSD_MAX_SWEEPS);
  return COSMO_HIP_OK;
}

bool psd_needs_sync(const cosmo_hip_handle* h) { return h->psd && !h->psd->large.empty() && !polar_has_large(h); }

int32_t psd_enqueue_project(cosmo_hip_handle* h, real* s, bool guard_b) {
  PsdPlan* p = h->psd;
  if (!p || p->cones.empty()) return COSMO_HIP_OK;
  const int guard = guard_b ? 1 : 0;
  prof_begin(h, KC_PSD);
  HIPCHK(h, hipMemsetAsync(p->rank, 0, p->cones.size() * sizeof(int), h->stream));
  if (!p->tiny.empty()) {
    const int n = (int)p->tiny.size();
    hipLaunchKernelGGL(k_psd_tiny, dim3((n + COSMO_BS / 64 - 1) / (COSMO_BS / 64)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, n,
                       p->d_tiny, p->d_cones, s, p->rank, p->flags, 0, 1.0, p->eigmin);
  }
  for (size_t gi = 0; gi < p->pj_groups.size(); ++gi) {
    const int n = (int)p->pj_groups[gi].size();
    const int* lst = p->d_pj_groups[gi];
    hipLaunchKernelGGL(k_psd_populate, dim3(8, n), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, lst, p->d_cones, s, p->G, p->cshift, 1.0, 0);
    switch (p->pj_waves[gi]) {
      case 16: hipLaunchKernelGGL((k_psd_jacobi_wg<16>), dim3(n), dim3(1024), 0, h->stream, h->ctl, guard, lst, p->d_cones, p->G, p->cshift, p->flags, p->tol_factor, p->dbg); break;
      default: hipLaunchKernelGGL((k_psd_jacobi_wg<4>), dim3(n), dim3(256), 0, h->stream, h->ctl, guard, lst, p->d_cones, p->G, p->cshift, p->flags, p->tol_factor, p->dbg); break;
    }
    hipLaunchKernelGGL(k_psd_colscale, dim3(4, n), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, lst, p->d_cones, p->G, p->cshift, p->colw, p->rank);
    int maxtiles = 1;
    for (int idx : p->pj_groups[gi]) { const int nt = p->cones[idx].ld / 16; maxtiles = std::max(maxtiles, nt * (nt + 1) / 2); }
    hipLaunchKernelGGL(k_psd_syrk, dim3((maxtiles + 3) / 4, n), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, lst, p->d_cones, p->G, s);
  }
  if (polar_has_batch(h)) CHK(polar_enqueue_project_batch(h, s, guard));
  if (polar_has_large(h)) CHK(polar_enqueue_project(h, s, guard));
  if (!p->large.empty() && !(polar_has_large(h) && p->large_by_polar)) {
    // host-paced: one launch per tournament step, convergence flag read once per sweep
    if (guard) {
      CHK(sync_ctl(h));
      if (h->ctl_host->halt) { prof_end(h); return COSMO_HIP_OK; }
    }
    const int n = (int)p->large.size();
    int nbmax = 0;
    for (int idx : p->large) nbmax = std::max(nbmax, p->cones[idx].nb);
    hipLaunchKernelGGL(k_psd_populate, dim3(64, n), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, p->d_large, p->d_cones, s, p->G, p->cshift, 1.0, 0);
    CHK(psd_large_sweeps(h, n, nbmax));
    hipLaunchKernelGGL(k_psd_colscale, dim3(128, n), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, p->d_large, p->d_cones, p->G, p->cshift, p->colw, p->rank);
    int maxtiles = 1;
    for (int idx : p->large) { const int nt = p->cones[idx].ld / 16; maxtiles = std::max(maxtiles, nt * (nt + 1) / 2); }
    hipLaunchKernelGGL(k_psd_syrk, dim3(std::min(8192, (maxtiles + 3) / 4), n), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, p->d_large, p->d_cones, p->G, s);
  }
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t psd_get_ranks(cosmo_hip_handle* h, int64_t* rank_per_cone) {
  PsdPlan* p = h->psd;
  if (!p || p->cones.empty()) return COSMO_HIP_OK;
  std::vector<int> r(p->cones.size());
  int fl[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpyAsync(r.data(), p->rank, r.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(fl, p->flags, sizeof fl, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (fl[1]) return cosmo_fail(h, COSMO_HIP_ERR_EIG, "Jacobi eigensolver did not converge");
  for (size_t i = 0; i < p->cones.size(); ++i) rank_per_cone[p->cones[i].cone_index] = r[i];
  return COSMO_HIP_OK;
}

// diagnostics: out = {max sweeps of the workgroup kernels, sweeps of the last multi-workgroup solve, error flag, #cones}
extern "C" int32_t cosmo_hip_psd_stats(cosmo_hip_handle* h, int64_t out[4]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  out[0] = out[1] = out[2] = out[3] = 0;
  PsdPlan* p = h->psd;
  if (!p || p->cones.empty()) return COSMO_HIP_OK;
  int fl[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpyAsync(fl, p->flags, sizeof fl, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out[0] = fl[2]; out[1] = p->last_large_sweeps; out[2] = fl[1]; out[3] = (int64_t)p->cones.size();
  return COSMO_HIP_OK;
}

// Smallest eigenvalue of sign * mat(vec slice) for every planned PSD cone (same row layout as s).  Used by the
// infeasibility certificates: is_pos_def!(X, tol) <=> lambda_min(X) > -tol (src/algebra.jl:226-238).  Synchronous.
int32_t polar_complex_is_pd(cosmo_hip_handle* h, const real* vec, real sign, real tol, std::vector<int>& ok);   // psd_polar.hip

int32_t psd_extreme_eigs(cosmo_hip_handle* h, const real* vec, real sign, real tol, std::vector<real>& lam_min) {
  PsdPlan* p = h->psd;
  lam_min.clear();
  if (!p || p->cones.empty()) return COSMO_HIP_OK;
  real* v = const_cast<real*>(vec);   // mode 1 / populate only read it
  if (!p->tiny.empty()) {
    const int n = (int)p->tiny.size();
    hipLaunchKernelGGL(k_psd_tiny, dim3((n + COSMO_BS / 64 - 1) / (COSMO_BS / 64)), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, n, p->d_tiny,
                       p->d_cones, v, p->rank, p->flags, 1, sign, p->eigmin);
  }
  for (size_t gi = 0; gi < p->wg_groups.size(); ++gi) {
    const int n = (int)p->wg_groups[gi].size();
    const int* lst = p->d_wg_groups[gi];
    hipLaunchKernelGGL(k_psd_populate, dim3(8, n), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, lst, p->d_cones, vec, p->G, p->cshift, sign, 1);
    if (p->wg_waves[gi] == 16)
      hipLaunchKernelGGL((k_psd_jacobi_wg<16>), dim3(n), dim3(1024), 0, h->stream, h->ctl, 0, lst, p->d_cones, p->G, p->cshift, p->flags, p->tol_factor, 0);
    else
      hipLaunchKernelGGL((k_psd_jacobi_wg<4>), dim3(n), dim3(256), 0, h->stream, h->ctl, 0, lst, p->d_cones, p->G, p->cshift, p->flags, p->tol_factor, 0);
    hipLaunchKernelGGL(k_psd_eigmin, dim3(n), dim3(COSMO_BS), 0, h->stream, lst, p->d_cones, p->G, p->cshift, p->eigmin);
  }
  if (!p->large.empty()) {
    const int n = (int)p->large.size();
    int nbmax = 0;
    for (int idx : p->large) nbmax = std::max(nbmax, p->cones[idx].nb);
    hipLaunchKernelGGL(k_psd_populate, dim3(64, n), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, p->d_large, p->d_cones, vec, p->G, p->cshift, sign, 1);
    CHK(psd_large_sweeps(h, n, nbmax));
    hipLaunchKernelGGL(k_psd_eigmin, dim3(n), dim3(COSMO_BS), 0, h->stream, p->d_large, p->d_cones, p->G, p->cshift, p->eigmin);
  }
  HIPCHK(h, hipGetLastError());
  lam_min.resize(p->cones.size());
  HIPCHK(h, hipMemcpyAsync(lam_min.data(), p->eigmin, sizeof(real) * lam_min.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // Hermitian cones: the reference's own test -- does the Cholesky factorisation of sign * H + tol I succeed? -- on the real embedding
  // (psd_polar.hip); reported as lambda_min = 0 (passes the caller's "> -tol") or -inf.  Sides 2r > 1024 never certify.
  if (!p->cplx.empty()) {
    std::vector<int> ok;
    CHK(polar_complex_is_pd(h, vec, sign, tol, ok));
    for (size_t q = 0; q < p->cplx.size(); ++q) lam_min[p->cplx[q]] = ok[q] ? 0.0 : -INFINITY;
  }
  return COSMO_HIP_OK;
}
