// psd16.h -- the wave-level PSD projection of cones with side d <= 16 (src/convexset.jl:219-263, 303-321, 402-412), shared by the single-problem path
// (psd.hip: k_psd_tiny, one wave per cone) and the batch kernels (batch.hip: the persistent workgroup of a problem projects its small PSD cones with
// its own waves).  One code path => the same bits in both.
#pragma once
#include "psd_internal.h"

// ---------------------------------------------------------------------------------------------------------------------
// 16x16 Jacobi machinery (one wave).  W and J live in LDS in row-major [16][WLD]; lane l works on column (l & 15) and
// rows (l >> 4) + 4 r, r = 0..3 -- the C/D layout of v_mfma_f64_16x16x4_f64, so MFMA results drop in without shuffles.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Rotation that annihilates w_pq:  t = sign(delta) w_pq / (|delta| + sqrt(delta^2 + w_pq^2)), delta = (w_qq - w_pp)/2.
// t only steers convergence, so it uses the hardware approximations (v_sqrt_f64, v_rcp_f64); c = (1+t^2)^(-1/2) is
// refined with one Newton step because c^2 + s^2 = 1 must hold to rounding for J to stay orthogonal.
__device__ __forceinline__ void jacobi_cs(real app, real aqq, real apq, real& c, real& s) {
  const real delta = R(0.5) * (aqq - app);
  const real h = hw_sqrt(fma(delta, delta, apq * apq));
  const real den = fabs(delta) + h;
  const real t = ((delta >= R(0.0)) ? apq : -apq) * hw_rcp(den);
  const real a = fma(t, t, 1.0);
  real y = hw_rsq(a);
  const real e = fma(-a * y, y, 1.0);          // 1 - a y^2
  y = fma(R(0.5) * e, y, y);
  c = y;
  s = y * t;
}

// One sweep of 8-rotation rounds on the 16x16 matrix W (LDS), accumulating J.
//  nrounds = 15: all 120 pairs (cyclic round-robin).  nrounds = 8: only the 64 cross pairs (p < 8 <= q) of a block pair.
//  mode 0: W is a Gram matrix (relative criterion |w_pq| > tol sqrt(w_pp w_qq); columns with w_kk <= tiny are skipped).
//  mode 1: W is the symmetric matrix itself (absolute criterion |w_pq| > tiny).
// Returns (wave-uniform) whether any rotation fired.
__device__ __forceinline__ int jacobi16_sweep(real* W, real* J, int* part, real* ca, real* cb, real tol,
                                              real tiny, int mode, int nrounds, int lane) {
  int rotated = 0;
  for (int rd = 0; rd < nrounds; ++rd) {
    if (lane < 8) {
      int p, q;
      if (nrounds == 8) { p = lane; q = 8 + ((lane + rd) & 7); }
      else if (lane == 0) { p = rd; q = 15; }
      else { p = (rd + lane) % 15; q = (rd - lane + 15) % 15; }
      if (p > q) { const int t = p; p = q; q = t; }
      const real app = W[p * WLD + p], aqq = W[q * WLD + q], apq = W[p * WLD + q];
      real c = 1.0, s = 0.0;
      bool rot;
      if (mode == 0) rot = (app > tiny) && (aqq > tiny) && (apq * apq > (tol * tol) * (app * aqq));
      else rot = fabs(apq) > tiny;
      if (rot) { jacobi_cs(app, aqq, apq, c, s); rotated = 1; }
      part[p] = q; ca[p] = c; cb[p] = -s;   // col_p' = c col_p - s col_q
      part[q] = p; ca[q] = c; cb[q] = s;    // col_q' = s col_p + c col_q
    }
    wave_lds_fence();
    const int j = lane & 15;
    const int pj = part[j];
    const real aj = ca[j], bj = cb[j];
    real wn[4], jn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ACC_ROW(lane, r);
      const int pi = part[i];
      const real ai = ca[i], bi = cb[i];
      const real wij = W[i * WLD + j], wipj = W[i * WLD + pj], wpij = W[pi * WLD + j], wpipj = W[pi * WLD + pj];
      wn[r] = ai * (aj * wij + bj * wipj) + bi * (aj * wpij + bj * wpipj);
      jn[r] = aj * J[i * WLD + j] + bj * J[i * WLD + pj];
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ACC_ROW(lane, r);
      W[i * WLD + j] = wn[r];
      J[i * WLD + j] = jn[r];
    }
    wave_lds_fence();
  }
  return __any(rotated) ? 1 : 0;
}

// svec index of (i, j), i <= j (0-based), column-major upper triangle
__device__ __forceinline__ long long svec_idx(int i, int j) { return (long long)j * (j + 1) / 2 + i; }

// LDS workspace of one wave: W and J (16 x WLD each), the rotation tables
#define PSD16_WS_REALS (2 * 16 * WLD + 2 * 16)
#define PSD16_WS_BYTES ((int)(PSD16_WS_REALS * sizeof(real) + 16 * sizeof(int)))
struct Psd16Ws { real* W; real* J; real* ca; real* cb; int* part; };
__device__ __forceinline__ Psd16Ws psd16_ws_at(unsigned char* base) {
  Psd16Ws w;
  w.W = reinterpret_cast<real*>(base); w.J = w.W + 16 * WLD; w.ca = w.J + 16 * WLD; w.cb = w.ca + 16;
  w.part = reinterpret_cast<int*>(w.cb + 16);
  return w;
}

// One wave, one cone: x = the cone's slice (svec with sqrt(2) off-diagonals, or the d x d square), d <= 16.
// mode 0: project in place, returns the rank nnz_lambda (#{lambda > 0}).  mode 1: x is only read; *eigmin_out = smallest eigenvalue of
// sign * mat(x) (definiteness tests of the infeasibility certificates, src/algebra.jl:226-238).  *nonconv is set when the sweeps did not converge.
__device__ __forceinline__ int psd16_wave(real* x, int d, int kind, const Psd16Ws& ws, int lane, int mode, real sign, real* eigmin_out, int* nonconv) {
  real* W = ws.W; real* J = ws.J;
  const real isq2 = 1.0 / sqrt(2.0), sq2 = sqrt(2.0);
  // load X (symmetric, zero padded) and the identity
  real fro = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ACC_ROW(lane, r), j = lane & 15;
    real v = 0.0;
    if (i < d && j < d) {
      const int a = i < j ? i : j, b = i < j ? j : i;
      if (kind == COSMO_HIP_PSD_TRIANGLE) {
        const real t = x[svec_idx(a, b)];
        v = (a == b) ? t : isq2 * t;                       // populate_upper_triangle! (convexset.jl:432-442)
      } else {
        v = (mode == 1) ? x[(long long)b * d + a]                         // is_pos_def!: Hermitian(X, 'U') as is (algebra.jl:226-233)
                        : (x[(long long)b * d + a] + x[(long long)a * d + b]) / R(2.0);   // symmetrize_upper! (algebra.jl:201-208)
      }
    }
    v = v * sign;
    W[i * WLD + j] = v;
    J[i * WLD + j] = (i == j) ? 1.0 : 0.0;
    fro += v * v;
  }
  fro = sqrt(wave_sum(fro));
  wave_lds_fence();
  const real thr = PSD_EPS * fro;
  int sweeps = 0, rot = (fro > R(0.0)) ? 1 : 0;
  while (rot && sweeps < PSD_MAX_SWEEPS) {
    rot = jacobi16_sweep(W, J, ws.part, ws.ca, ws.cb, 0.0, thr, 1, 15, lane);
    ++sweeps;
  }
  if (rot && nonconv) *nonconv = 1;
  if (mode == 1) {
    real lm = W[0];
    for (int k = 1; k < d; ++k) lm = fmin(lm, W[k * WLD + k]);
    if (eigmin_out) *eigmin_out = lm;
    return 0;
  }
  // X+ = J max(Lambda,0) J'
  int rk = 0;
  for (int k = 0; k < d; ++k) rk += (W[k * WLD + k] > R(0.0)) ? 1 : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ACC_ROW(lane, r), j = lane & 15;
    if (i <= j && j < d) {
      real acc = 0.0;
      for (int k = 0; k < d; ++k) {
        const real lam = W[k * WLD + k];
        if (lam > R(0.0)) acc += (J[i * WLD + k] * lam) * J[j * WLD + k];
      }
      if (kind == COSMO_HIP_PSD_TRIANGLE) {
        x[svec_idx(i, j)] = (i == j) ? acc : sq2 * acc;              // extract_upper_triangle! (:462-472)
      } else {
        x[(long long)j * d + i] = acc;                                // upper triangle, then mirrored (:316-318)
        x[(long long)i * d + j] = acc;
      }
    }
  }
  return rk;
}
