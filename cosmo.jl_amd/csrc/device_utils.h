// device_utils.h -- wave64 / workgroup reductions and the CSR-stream row-block primitive (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "internal.h"

// partial-reduction slots (h->partials + slot * COSMO_MAX_PARTIALS)
enum { SLOT_BB = 0, SLOT_RR, SLOT_UC, SLOT_RP, SLOT_MP, SLOT_RD, SLOT_MD, SLOT_XPX, SLOT_QX, SLOT_AUX0, SLOT_AUX1, SLOT_AUX2 };
#define COSMO_NSLOTS_TOTAL 12

struct CsrView {
  int xcd_affine;    // 1: workgroup b of a one-tile-per-workgroup launch takes tile (b % 8) * (nb / 8) + b / 8 (see tile_of_block)
  const int* rowptr;
  const int* col;
  const real* val;
  const int* split;  // may be null
  const int* rb;     // 4 ints per tile: {r0, r1, nz0, nz1}
  int nb;
  int nrows;
  int split_col;     // columns >= split_col gather from x2[col - split_col]
};

static inline CsrView view_of(const CsrDev& D) {
  CsrView v;
  v.rowptr = D.rowptr; v.col = D.col; v.val = D.val; v.split = D.split; v.rb = D.rb;
  v.nb = D.nb; v.nrows = D.nrows; v.split_col = D.split_col;
  v.xcd_affine = (D.xcd_affine && D.grid == D.nb) ? 1 : 0;
  return v;
}

// XCD-affine tile order.  Workgroup b of a launch is observed on XCD b % 8 and every XCD has its own L2; with the identity order the
// eight L2s each see every eighth tile, i.e. rows from all over the matrix.  Here XCD x works through ONE contiguous range of tiles
// (rows), so that the lines of the row-indexed operands and -- for matrices with column locality -- of the gathered vector are
// fetched by one L2 instead of eight.  Only a speed choice: results (including the per-tile partials, which are indexed by TILE)
// do not depend on it.  Tiles beyond 8 * (nb / 8) keep the identity order.
__device__ __forceinline__ int tile_of_block(int b, int nb, int affine) {
  const int per = nb >> 3;
  return (affine && b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;
}

// Butterfly reductions (xor 32, 16, 8, 4, 2, 1): every lane ends with the same value, combination order is fixed => deterministic.
// No step goes through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to): the steps across rows of 16 lanes use gfx950's
// v_permlane32_swap / v_permlane16_swap (with both operands = v they leave {v[i], v[i ^ 32]} resp. {v[i], v[i ^ 16]} in the two result
// registers of every lane -- in the upper half / the odd rows in the opposite order, which an addition does not see), the four steps inside
// a row are DPP moves with the SAME partners as the xor butterfly: row_ror:8 reads lane i ^ 8; row_ror:4 reads lane (i + 4) mod 16, which
// after the xor-8 step holds the value of lane i ^ 4 (the row is 8-periodic by then); quad_perm [2,3,0,1] and [1,0,3,2] are xor 2 and
// xor 1.  Same additions in the same order as six __shfl_xor steps (bit-identical results: bench/wave_reduce_lab.hip) without the six
// crossbar round trips -- the reductions sit on the critical path of every latency-bound Krylov kernel and of the batch kernel's CG loop.
// Callers keep all 64 lanes of the wave active (every call site is wave-uniform): a swap or DPP move out of an inactive lane is not a zero.
template <int CTRL>
__device__ __forceinline__ real dpp_move(real v) {
#if REAL_IS_FLOAT
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
#else
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
#endif
}
// a, b = {v[i], v[i ^ (ROWS16 ? 16 : 32)]} in some order
template <bool ROWS16>
__device__ __forceinline__ void permlane_pair(real v, real& a, real& b) {
#if REAL_IS_FLOAT
  const int w = __float_as_int(v);
  const auto r = ROWS16 ? __builtin_amdgcn_permlane16_swap(w, w, false, false) : __builtin_amdgcn_permlane32_swap(w, w, false, false);
  a = __int_as_float((int)r[0]); b = __int_as_float((int)r[1]);
#else
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto r = ROWS16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto q = ROWS16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double((int)q[0], (int)r[0]); b = __hiloint2double((int)q[1], (int)r[1]);
#endif
}
#define DPP_ROW_ROR8 0x128
#define DPP_ROW_ROR4 0x124
#define DPP_QUAD_XOR2 0x4E
#define DPP_QUAD_XOR1 0xB1
__device__ __forceinline__ real wave_sum(real v) {
  real a, b;
  permlane_pair<false>(v, a, b); v = a + b;
  permlane_pair<true>(v, a, b); v = a + b;
  v += dpp_move<DPP_ROW_ROR8>(v);
  v += dpp_move<DPP_ROW_ROR4>(v);
  v += dpp_move<DPP_QUAD_XOR2>(v);
  v += dpp_move<DPP_QUAD_XOR1>(v);
  return v;
}
__device__ __forceinline__ real wave_max(real v) {
  real a, b, t;
  // every step keeps a NaN of EITHER partner (Julia's norm(x, Inf) returns NaN when x holds one, residuals.jl:30-53): `t > v` alone
  // drops a NaN held by t; a NaN held by v survives because both comparisons are false
  permlane_pair<false>(v, a, b); v = (b > a || b != b) ? b : a;
  permlane_pair<true>(v, a, b); v = (b > a || b != b) ? b : a;
  t = dpp_move<DPP_ROW_ROR8>(v); v = (t > v || t != t) ? t : v;
  t = dpp_move<DPP_ROW_ROR4>(v); v = (t > v || t != t) ? t : v;
  t = dpp_move<DPP_QUAD_XOR2>(v); v = (t > v || t != t) ? t : v;
  t = dpp_move<DPP_QUAD_XOR1>(v); v = (t > v || t != t) ? t : v;
  return v;
}

// red: COSMO_BS/64 doubles of LDS.  Result is broadcast to all threads.
__device__ __forceinline__ real block_sum(real v, real* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  real t = 0.0;
#pragma unroll
  for (int i = 0; i < COSMO_BS / 64; ++i) t += red[i];
  return t;
}
// two sums behind ONE pair of barriers (red2: 2 * COSMO_BS/64 reals); each is added exactly as block_sum adds it (same bits)
__device__ __forceinline__ void block_sum2(real& a, real& b, real* red2) {
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red2[threadIdx.x >> 6] = a; red2[COSMO_BS / 64 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  real t = 0.0, u = 0.0;
#pragma unroll
  for (int i = 0; i < COSMO_BS / 64; ++i) { t += red2[i]; u += red2[COSMO_BS / 64 + i]; }
  a = t; b = u;
}
__device__ __forceinline__ real block_max(real v, real* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  real t = red[0];
#pragma unroll
  for (int i = 1; i < COSMO_BS / 64; ++i) t = (red[i] > t || red[i] != red[i]) ? red[i] : t;   // a NaN of any wave survives
  return t;
}
// Every workgroup re-reduces the (<= COSMO_MAX_PARTIALS) partials of the producing kernel in the same fixed
// order, so all workgroups of all consumer kernels see bit-identical scalars without a finalize launch.
// A thread owns the partials t, t + 256, ... (at most COSMO_MAX_PARTIALS / COSMO_BS = 8) and adds them in that order.  All of its loads
// are REQUESTED BEFORE THE FIRST ADDITION: written as `for (i = t; i < count; i += 256) a += p[i]` the compiler emits load, s_waitcnt
// vmcnt(0), add per iteration -- up to eight serial L2 round trips at the head of every latency-bound Krylov kernel (k_cg_upd on
// BASELINE config 2 reduces 2048 partials: 4 of its 5.3 us).  Missing slots contribute +0.0, which cannot change the sum: the running
// sum starts at +0.0 and therefore is never -0.0.
#define COSMO_PARTS_PER_THREAD (COSMO_MAX_PARTIALS / COSMO_BS)
__device__ __forceinline__ real partials_prefetch_sum(const real* p, int count) {
  real v[COSMO_PARTS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < COSMO_PARTS_PER_THREAD; ++k) { const int i = (int)threadIdx.x + k * COSMO_BS; v[k] = (i < count) ? p[i] : R(0.0); }
  real a = 0.0;
#pragma unroll
  for (int k = 0; k < COSMO_PARTS_PER_THREAD; ++k) a += v[k];
  return a;
}
__device__ __forceinline__ real reduce_partials_sum(const real* p, int count, real* red) { return block_sum(partials_prefetch_sum(p, count), red); }
__device__ __forceinline__ real reduce_partials_max(const real* p, int count, real* red) {
  real v[COSMO_PARTS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < COSMO_PARTS_PER_THREAD; ++k) { const int i = (int)threadIdx.x + k * COSMO_BS; v[k] = (i < count) ? p[i] : R(0.0); }
  real a = 0.0;
#pragma unroll
  for (int k = 0; k < COSMO_PARTS_PER_THREAD; ++k) { const real t = v[k]; a = (t > a || t != t) ? t : a; }   // a padding 0.0 never replaces a
  return block_max(a, red);
}

// min(max(v, lo), hi) as Julia evaluates it (parameters.jl:65): a NaN stays a NaN (C's fmax / fmin return the OTHER operand), so that a NaN
// residual ratio fails both comparisons of the adaptive-rho rule and leaves rho alone, as in the reference
__device__ __forceinline__ real clamp_keep_nan(real v, real lo, real hi) { return (v != v) ? v : fmin(fmax(v, lo), hi); }

// abs-max that propagates NaN like Julia's norm(x, Inf)
__device__ __forceinline__ real amax(real acc, real v) {
  real a = fabs(v);
  return (a > acc || a != a) ? a : acc;
}

// CSR-stream over one row block [r0, r1): phase 1 stages val*x products of the block's contiguous nonzero range
// in LDS with fully coalesced loads of (val, col); phase 2 gives every row to one thread, which adds its LDS
// segment left to right -- exactly the order in which Julia's CSC kernels accumulate (no FMA contraction), so the
// row sums are bit-identical to the serial CPU loop.  Rows longer than the LDS tile take the chunked path.
// fn(row, sum1, sum2): sum1 over [rowptr[row], split[row]) and sum2 over [split[row], rowptr[row+1]).
// Left-to-right sum of lds[a .. b): the additions happen strictly in index order (the order of Julia's CSC kernels: bit-identical row
// sums), but four LDS reads are requested before the four dependent additions -- a 20-nonzero row (A', [P | A']) otherwise pays one
// LDS round trip per nonzero with only a third of the workgroup's threads owning a row.
__device__ __forceinline__ real lds_seq_sum(const real* lds, int a, int b) {
  real s = 0.0;
  int k = a;
  // one ds_read_b64 per element: the compiler pairs neighbouring reads into ds_read2_b64 / ds_read_b128, which gfx950 serves at half the rate of single b64 reads
  // (a volatile access in the LDS address space is not merged); cfg2 k_op_apply 17.3 -> 16.9 us, same sums
  const volatile __attribute__((address_space(3))) real* l3 = (const volatile __attribute__((address_space(3))) real*)lds;
  for (; k + 4 <= b; k += 4) {
    const real v0 = l3[k], v1 = l3[k + 1], v2 = l3[k + 2], v3 = l3[k + 3];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; k < b; ++k) s += l3[k];
  return s;
}

// gat(c): the operand gathered for column c (the plain form below reads x1 / x2; the fused direction + product kernel of the CG
// iteration rebuilds u = r + beta u_old at the gathered column).
template <class GatherFn, class RowFn>
__device__ __forceinline__ void csr_stream_rows_g(const CsrView& M, GatherFn gat, int r0, int r1, int nz0, int nz1, real* lds,
                                                  real* red, RowFn fn) {
  const int cnt = nz1 - nz0;
  if (cnt <= COSMO_NNZ_PER_BLOCK) {
    // row pointers of this thread's first row: requested together with (col, val) so that the dependent chain of these 3-20 us
    // kernels is descriptor -> {col, val, rowptr} -> gather instead of descriptor -> {col, val} -> gather -> rowptr
    const int rfirst = r0 + threadIdx.x;
    int pa = 0, pb = 0, psp = 0;
    if (rfirst < r1) { pa = M.rowptr[rfirst]; pb = M.rowptr[rfirst + 1]; psp = M.split ? M.split[rfirst] : pb; }
#pragma unroll
    for (int it = 0; it < COSMO_NNZ_PER_BLOCK / COSMO_BS; ++it) {
      const int k = it * COSMO_BS + threadIdx.x;
      if (k < cnt) {
        const int c = M.col[nz0 + k];
        const real a = M.val[nz0 + k];
        const real xv = gat(c);
        lds[k] = a * xv;
      }
    }
    __syncthreads();
    for (int r = rfirst; r < r1; r += COSMO_BS) {
      const int a = ((r == rfirst) ? pa : M.rowptr[r]) - nz0;
      const int b = ((r == rfirst) ? pb : M.rowptr[r + 1]) - nz0;
      const int sp = (r == rfirst) ? (psp - nz0) : (M.split ? (M.split[r] - nz0) : b);
      const real s1 = lds_seq_sum(lds, a, sp);
      const real s2 = lds_seq_sum(lds, sp, b);
      fn(r, s1, s2);
    }
    __syncthreads();
  } else {
    // single long row (the schedule never mixes a long row with others)
    const int r = r0;
    const int sp = M.split ? M.split[r] : nz1;
    real s1 = 0.0, s2 = 0.0;
    for (int k = nz0 + threadIdx.x; k < nz1; k += COSMO_BS) {
      const int c = M.col[k];
      const real xv = gat(c);
      const real p = M.val[k] * xv;
      if (k < sp) s1 += p; else s2 += p;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) fn(r, s1, s2);
    __syncthreads();
  }
}

template <class RowFn>
__device__ __forceinline__ void csr_stream_rows(const CsrView& M, const real* __restrict__ x1,
                                                const real* __restrict__ x2, int r0, int r1, int nz0, int nz1, real* lds,
                                                real* red, RowFn fn) {
  const int split_col = M.split_col;
  csr_stream_rows_g(M, [&](int c) { return (c < split_col) ? x1[c] : x2[c - split_col]; }, r0, r1, nz0, nz1, lds, red, fn);
}

// Tile k of the CSR-stream schedule.  M.rb holds one 16-byte descriptor {first row, end row, first nonzero, end nonzero}
// per tile, so a workgroup needs ONE dependent load (instead of rb[k], rb[k+1], rowptr[r0], rowptr[r1]) before it can
// start streaming -- the ramp-up of these 10-20 us kernels is a chain of dependent memory round trips.
template <class RowFn>
__device__ __forceinline__ void csr_stream_tile(const CsrView& M, const real* __restrict__ x1, const real* __restrict__ x2,
                                                int k, real* lds, real* red, RowFn fn) {
  const int4 d = reinterpret_cast<const int4*>(M.rb)[k];
  csr_stream_rows(M, x1, x2, d.x, d.y, d.z, d.w, lds, red, fn);
}
