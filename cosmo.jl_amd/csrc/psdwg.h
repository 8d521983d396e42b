// psdwg.h -- workgroup-level PSD projection of ONE cone with 16 < d <= 256 by block one-sided (Hestenes) Jacobi on G = X + c I with the Gram /
// update products on the matrix cores (src/convexset.jl:219-263: X+ = sum_{lambda > 0} lambda z z'; the design is described at the top of psd.hip).
// Shared by the single-problem path (psd.hip: k_psd_jacobi_wg, one workgroup per cone) and -- round 4 -- by the persistent workgroups of the batch
// kernels (batch.hip: PSD cones of side 17 .. 64 of a batch of small SDPs), which run populate -> sweeps -> column scaling -> SYRK for one cone after
// the other with all their waves.
#pragma once
#include "psd16.h"

// Does any pair of the panel need a rotation?  cross_only: test the 64 entries (p < 8 <= q), one per lane; otherwise all
// 120 pairs (two per lane).  Wave-uniform result; lets converged block pairs skip the Jacobi sweep and the panel update.
__device__ __forceinline__ int gram_needs_work(const real* W, real tol, real tiny, int cross_only, int lane) {
  int need = 0;
  if (cross_only) {
    const int p = lane & 7, q = 8 + (lane >> 3);
    const real app = W[p * WLD + p], aqq = W[q * WLD + q], apq = W[p * WLD + q];
    need = (app > tiny) && (aqq > tiny) && (apq * apq > (tol * tol) * (app * aqq));
  } else {
    for (int e = lane; e < 120; e += 64) {
      int p = 0, rem = e;
      while (rem >= 15 - p) { rem -= 15 - p; ++p; }
      const int q = p + 1 + rem;
      const real app = W[p * WLD + p], aqq = W[q * WLD + q], apq = W[p * WLD + q];
      need |= (app > tiny) && (aqq > tiny) && (apq * apq > (tol * tol) * (app * aqq));
    }
  }
  return __any(need) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// block-pair step pieces (wave level).  cols[0..15] = global column indices of the 16 panel columns.
// ---------------------------------------------------------------------------------------------------------------------
// Gram matrix of a row range of the panel: W = P(r0:r1, :)' P(r0:r1, :), r0, r1 multiples of 16.
// Lane l loads rows 16 ch + 4 (l >> 4) .. +3 of column cols(l & 15) as one 32-byte vector; the k-slot permutation this
// implies is the same for both MFMA operands, so the sum is unchanged.
__device__ __forceinline__ v4d panel_gram(const real* __restrict__ g, int ld, int colL, int r0, int r1, int lane) {
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  const real* cp = g + (long long)colL * ld + 4 * (lane >> 4);
  int r = r0;
  for (; r + 64 <= r1; r += 64) {                      // four 16-row chunks in flight
    v4d v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4d*>(cp + r + 16 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = MFMA_REAL(v[u].x, v[u].x, acc);
      acc = MFMA_REAL(v[u].y, v[u].y, acc);
      acc = MFMA_REAL(v[u].z, v[u].z, acc);
      acc = MFMA_REAL(v[u].w, v[u].w, acc);
    }
  }
  for (; r < r1; r += 16) {
    const v4d v = *reinterpret_cast<const v4d*>(cp + r);
    acc = MFMA_REAL(v.x, v.x, acc);
    acc = MFMA_REAL(v.y, v.y, acc);
    acc = MFMA_REAL(v.z, v.z, acc);
    acc = MFMA_REAL(v.w, v.w, acc);
  }
  return acc;
}

// Panel update P(r0:r1, :) <- P(r0:r1, :) J, computed as (J' P')' so that every lane stores 16 consecutive rows of one
// column.  jt[t] = J[(lane >> 4) + 4 t][lane & 15] (the C layout) is exactly the A operand of step t.
__device__ __forceinline__ void panel_update(real* __restrict__ g, int ld, const int* cols, const real jt[4], int r0, int r1,
                                             int lane) {
  const int rr = lane & 15, kg = lane >> 4;
  real* src[4];
  real* dst[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    src[t] = g + (long long)cols[4 * t + kg] * ld + rr;   // B operand of step t: P[r + rr][4 t + kg]
    dst[t] = g + (long long)cols[ACC_ROW(lane, t)] * ld + rr;   // D reg t: new column ACC_ROW(lane, t) (kg + 4 t in fp64), row r + rr
  }
  int r = r0;
  for (; r + 32 <= r1; r += 32) {                      // two chunks in flight (all loads of both chunks precede the stores)
    real b0[4], b1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { b0[t] = src[t][r]; b1[t] = src[t][r + 16]; }
    v4d a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) { a0 = MFMA_REAL(jt[t], b0[t], a0); a1 = MFMA_REAL(jt[t], b1[t], a1); }
    dst[0][r] = a0.x; dst[1][r] = a0.y; dst[2][r] = a0.z; dst[3][r] = a0.w;
    dst[0][r + 16] = a1.x; dst[1][r + 16] = a1.y; dst[2][r + 16] = a1.z; dst[3][r + 16] = a1.w;
  }
  for (; r < r1; r += 16) {
    real b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = src[t][r];
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = MFMA_REAL(jt[t], b[t], acc);
    dst[0][r] = acc.x; dst[1][r] = acc.y; dst[2][r] = acc.z; dst[3][r] = acc.w;
  }
}

// round-robin tournament: block pair w (0 <= w < nb/2) of step st (0 <= st < nb-1), nb even
__device__ __forceinline__ void rr_pair(int nb, int st, int w, int& I, int& J) {
  const int m = nb - 1;
  if (w == 0) { I = st % m; J = m; }
  else { I = (st + w) % m; J = (st - w + m) % m; }
  if (I > J) { const int t = I; I = J; J = t; }
}


// per-wave LDS workspace of the block-pair visits: the 16 x 16 Gram matrix W and the rotation accumulator J (as psd16.h) + the panel's column list
#define PSDWG_WS_STRIDE ((PSD16_WS_BYTES + 16 * (int)sizeof(int) + 15) / 16 * 16)
struct PsdWgWs { Psd16Ws w; int* cols; };
__device__ __forceinline__ PsdWgWs psdwg_ws_at(unsigned char* base) {
  PsdWgWs q;
  q.w = psd16_ws_at(base);
  q.cols = q.w.part + 16;
  return q;
}

// The whole Jacobi process of one cone by the NW waves of a workgroup (nb / 2 <= NW block pairs per tournament step): sweeps until no rotation
// fires.  g: ld x (8 nb) column-major, c = the shift (||X||_F), any_rot: one shared int.  Every thread of the workgroup must call it (barriers).
// Returns the number of sweeps executed (PSD_MAX_SWEEPS = not converged).
template <int NW>
__device__ __forceinline__ int psdwg_jacobi(real* __restrict__ g, int ld, int nb, int d, real c, real tolf, int dbg, unsigned char* ws_base, int* any_rot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int npairs = nb / 2;
  const real tol = tolf * (real)d * PSD_EPS;
  const real tiny = (tol * c) * (tol * c);
  const PsdWgWs ws = psdwg_ws_at(ws_base + (size_t)(wv < NW ? wv : 0) * PSDWG_WS_STRIDE);
  real* W = ws.w.W; real* J = ws.w.J;
  int sweep = 0;
  // one visit of block pair (I, Jb): Gram on MFMA, Jacobi on the 16x16 Gram matrix, panel update on MFMA
  auto visit = [&](int I, int Jb, int full) {
    if (lane < 16) ws.cols[lane] = (lane < 8) ? (I * 8 + lane) : (Jb * 8 + lane - 8);
    wave_lds_fence();
    const int colL = ws.cols[lane & 15];
    v4d w = {1.0, 0.5, 0.25, 0.125};
    if (!(dbg & 4)) w = panel_gram(g, ld, colL, 0, ld, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ACC_ROW(lane, r), j = lane & 15;
      W[i * WLD + j] = w[r];
      J[i * WLD + j] = (i == j) ? 1.0 : 0.0;
    }
    wave_lds_fence();
    if (!(dbg & 8) && !gram_needs_work(W, tol, tiny, !full, lane)) return;
    int rot = 1;
    if (!(dbg & 1)) rot = jacobi16_sweep(W, J, ws.w.part, ws.w.ca, ws.w.cb, tol, tiny, 0, full ? 15 : 8, lane);
    if (dbg & 2) rot = 0;
    if (dbg & 16) { if (lane == 0 && sweep < 10) *any_rot = 1; }
    if (rot) {
      real jt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) jt[t] = J[((lane >> 4) + 4 * t) * WLD + (lane & 15)];
      panel_update(g, ld, ws.cols, jt, 0, ld, lane);
      if (lane == 0) *any_rot = 1;
    }
  };
  for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
    if (threadIdx.x == 0) *any_rot = 0;
    __syncthreads();
    // diagonal pass: all pairs inside blocks (2w, 2w+1); the tournament steps then only rotate cross pairs
    if (wv < npairs && wv < NW) visit(2 * wv, 2 * wv + 1, 1);
    __syncthreads();
    for (int st = 0; st < nb - 1; ++st) {
      if (wv < npairs && wv < NW) {
        int I, Jb;
        rr_pair(nb, st, wv, I, Jb);
        visit(I, Jb, 0);
      }
      __syncthreads();
    }
    if (!*any_rot) break;
    __syncthreads();
  }
  return sweep;
}

// ---- the other three phases for ONE cone handled by a whole workgroup of BS threads (batch kernels) -----------------------------------------------
// ||X||_F and G = sign * X + c I (zero padded to ld x ncp); x = the cone's slice (svec / square layout; global or LDS).  Returns c.
// upper_only: Hermitian(X, 'U') of the square layout as is (definiteness tests), otherwise symmetrised like project! does.
template <int BS>
__device__ __forceinline__ real psdwg_populate(const real* x, int d, int kind, int ld, int ncp, real sign, int upper_only, real* __restrict__ g, real* red) {
  const real isq2 = 1.0 / sqrt(2.0);
  real acc = 0.0;
  if (kind == COSMO_HIP_PSD_TRIANGLE) {
    const int len = d * (d + 1) / 2;
    for (int k = threadIdx.x; k < len; k += BS) { const real v = x[k]; acc += v * v; }
  } else {
    for (int k = threadIdx.x; k < d * d; k += BS) {
      const int i = k % d, j = k / d;
      const int a = i < j ? i : j, b = i < j ? j : i;
      const real v = upper_only ? x[b * d + a] : (x[j * d + i] + x[i * d + j]) / R(2.0);
      acc += v * v;
    }
  }
  // block sum over BS threads (wave butterfly, then the wave sums in order)
  acc = wave_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  real tot = 0.0;
  for (int i = 0; i < BS / 64; ++i) tot += red[i];
  const real c = sqrt(tot);
  for (int e = threadIdx.x; e < ld * ncp; e += BS) {
    const int i = e % ld, j = e / ld;
    real v = 0.0;
    if (i < d && j < d) {
      const int a = i < j ? i : j, b = i < j ? j : i;
      if (kind == COSMO_HIP_PSD_TRIANGLE) {
        const real t = x[svec_idx(a, b)];
        v = (a == b) ? t : isq2 * t;
      } else {
        v = upper_only ? x[b * d + a] : (x[b * d + a] + x[a * d + b]) / R(2.0);
      }
      v = v * sign;
      if (i == j) v += c;
    }
    g[(long long)j * ld + i] = v;
  }
  __syncthreads();
  return c;
}

// sigma_k = ||g_k||, lambda_k = sigma_k - c; mode 0: ghat_k = g_k sqrt(lambda_k) / sigma_k for lambda_k > 0 else 0 (rank_k_update!, convexset.jl:248-256),
// then X+ = Ghat Ghat' (upper 16 x 16 tiles on the matrix cores, one wave per tile) written into x in the cone's layout.  mode 1: returns the smallest
// eigenvalue min_k (sigma_k - c) over the real columns (every thread gets it; x is not touched).
template <int BS>
__device__ __forceinline__ real psdwg_finish(real* x, int d, int kind, int ld, int ncp, real c, real* __restrict__ g, int mode, real* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  real lm = INFINITY;
  for (int j = wv; j < ncp; j += BS / 64) {
    real* col = g + (long long)j * ld;
    real a = 0.0;
    for (int i = lane; i < ld; i += 64) { const real v = col[i]; a += v * v; }
    const real sig = sqrt(wave_sum(a));
    const real lam = sig - c;
    if (mode == 1) { if (j < d) lm = fmin(lm, lam); continue; }
    real f = 0.0;
    if (j < d && lam > R(0.0) && sig > R(0.0)) f = sqrt(lam) / sig;
    for (int i = lane; i < ld; i += 64) col[i] = col[i] * f;
  }
  if (mode == 1) {
    __syncthreads();
    if (lane == 0) red[wv] = lm;
    __syncthreads();
    real m = red[0];
    for (int i = 1; i < BS / 64; ++i) m = fmin(m, red[i]);
    __syncthreads();
    return m;
  }
  __syncthreads();
  const int nt = ld / 16, ntiles = nt * (nt + 1) / 2;
  const real sq2 = sqrt(2.0);
  for (int t = wv; t < ntiles; t += BS / 64) {
    int tj = 0;
    while ((tj + 1) * (tj + 2) / 2 <= t) ++tj;
    const int ti = t - tj * (tj + 1) / 2;
    const real* pa = g + 16 * tj + (lane & 15) + (long long)(lane >> 4) * ld;
    const real* pb = g + 16 * ti + (lane & 15) + (long long)(lane >> 4) * ld;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < ncp; k += 4) {
      const real a = pa[(long long)k * ld];
      const real b = pb[(long long)k * ld];
      acc = MFMA_REAL(a, b, acc);
    }
    const int i = 16 * ti + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * tj + ACC_ROW(lane, r);
      if (i < d && j < d && i <= j) {
        if (kind == COSMO_HIP_PSD_TRIANGLE) x[svec_idx(i, j)] = (i == j) ? acc[r] : sq2 * acc[r];
        else { x[j * d + i] = acc[r]; x[i * d + j] = acc[r]; }
      }
    }
  }
  __syncthreads();
  return 0.0;
}
