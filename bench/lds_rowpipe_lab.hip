// lds_rowpipe_lab.hip -- how many nonzeros must a wave keep in flight in the sparse passes of the register batch kernel (k_batch_admm_reg)?
// One workgroup of 512 threads holds a config-3 sized problem in LDS exactly as the kernel does (values 80 KB, u16 column positions, packed u32
// (value position | row << 16) pairs of A', gathered vectors xv / tv) and runs the two passes of a Krylov iteration `reps` times:
//     A pass : thread t computes two rows (sorted positions t and 1023 - t, or 2t and 2t + 1), writes rho .* (A u) to tv, barrier
//     A' pass: thread t computes column t, barrier
// with the row loops in several forms (same products added in the same order: the checksums must agree bit for bit):
//   A pass   0  row_pipe3, the rows one after the other (the kernel today: one nonzero = 3 LDS reads in flight per wave)
//            1  both rows in lockstep (two nonzeros in flight), the shorter row's loads end at the wave's longest short row; then the long row alone
//            2  lockstep with the pairing (2t, 2t + 1): both rows of a thread have nearly the same length, waves are unbalanced
//   A' pass  0  row_pipe3
//            1  two nonzeros per trip (index pair with one ds_read2_b32, four operand loads issued together)
//            2  four nonzeros per trip
//            3 / 4  lean: packed entry = value ADDRESS | row << 21 (one VALU op per address), immediate offsets, scalar loop control; 1 / 2 nonzeros per trip
//   4 / 5  lean + sliced layout (entry (t, lane) of a wave-slot at t * 64 + lane, rows padded to the slot's longest);  6 / 7  lean + jagged layout (JDS, no padding)
//   A pass   3  lean: u16 entries are byte offsets of the gathered operand (the loaded index is the address), immediate offsets, scalar loop control
// Cycles: clock64 of thread 0 around each pass incl. its barrier, averaged over the repetitions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off bench/lds_rowpipe_lab.hip -o bench/lds_rowpipe_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>

typedef double real;
#define BS 512
#define RSH 3

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ void lds_read64(real& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_read_u32(uint32_t& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_read_u16(uint32_t& d, uint32_t addr) { asm volatile("ds_read_u16 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_read2_u32(uint32_t& d0, uint32_t& d1, uint32_t addr) {
  uint64_t d;
  asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d) : "v"(addr));
  d0 = (uint32_t)d; d1 = (uint32_t)(d >> 32);
}
__device__ __forceinline__ void lds_wait1(uint32_t& i, real& a, real& g) { asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(i), "+v"(a), "+v"(g)); }
__device__ __forceinline__ void lds_wait1(uint32_t& i) { asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(i)); }
__device__ __forceinline__ void lds_wait0(real& a, real& g) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(g)); }

// ---- form 0: the kernel's loop (csrc/batch.hip, row_pipe3) -------------------------------------------------------------------------------------
template <bool PAIR>
__device__ __forceinline__ real row_pipe3(uint32_t idx, uint32_t val, const uint32_t gat, const int len) {
  if (len <= 0) return 0.0;
  constexpr uint32_t ISZ = PAIR ? 4u : 2u;
  const uint32_t last = idx + ISZ * (uint32_t)(len - 1);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  auto load_idx = [&](uint32_t& d) { if (PAIR) lds_read_u32(d, idx); else lds_read_u16(d, idx); idx = (idx + ISZ < last) ? idx + ISZ : last; };
  auto load_ag = [&](real& a, real& g, uint32_t i, uint32_t vaddr) {
    if (PAIR) { lds_read64(a, val + ((i & 0xffffu) << RSH)); lds_read64(g, gat + ((i >> 16) << RSH)); }
    else { lds_read64(a, vaddr); lds_read64(g, gat + (i << RSH)); }
  };
  const uint32_t vlast = val + ((uint32_t)(len - 1) << RSH);
  uint32_t v1 = (len > 1) ? val + (1u << RSH) : vlast;
  load_idx(iA);
  load_idx(iB);
  lds_wait1(iA);
  load_ag(aA, gA, iA, val);
  int k = 0;
  for (;;) {
    load_idx(iA);
    lds_wait1(iB, aA, gA);
    load_ag(aB, gB, iB, v1);
    asm volatile("" : "+v"(aA), "+v"(gA));
    v1 = (v1 + (1u << RSH) < vlast) ? v1 + (1u << RSH) : vlast;
    s += aA * gA;
    if (++k >= len) break;
    load_idx(iB);
    lds_wait1(iA, aB, gB);
    load_ag(aA, gA, iA, v1);
    asm volatile("" : "+v"(aB), "+v"(gB));
    v1 = (v1 + (1u << RSH) < vlast) ? v1 + (1u << RSH) : vlast;
    s += aB * gB;
    if (++k >= len) break;
  }
  lds_wait0(aA, gA);
  return s;
}

// ---- A rows, two streams in lockstep ---------------------------------------------------------------------------------------------------------
// stream s: u16 positions at i_s, values at v_s, len_s nonzeros (per lane); L1 = wave maximum of len1, L0 = wave maximum of max(len0, len1) (uniform).
// Trips 0 .. L1-1 carry both streams (two index loads, wait for all but those, four operand loads, two products), trips L1 .. L0-1 stream 0 alone.
// Pointers are clamped to the row's last entry; a product past a row's end is not added.
struct StreamA {
  uint32_t ip, ilast, vp, vlast;
  __device__ __forceinline__ void init(uint32_t i, uint32_t v, int len) {
    const int l1 = len > 0 ? len - 1 : 0;
    ip = i; ilast = i + 2u * (uint32_t)l1; vp = v; vlast = v + ((uint32_t)l1 << RSH);
  }
  __device__ __forceinline__ void load_idx(uint32_t& d) { lds_read_u16(d, ip); ip = (ip + 2u < ilast) ? ip + 2u : ilast; }
  __device__ __forceinline__ void load_ag(real& a, real& g, uint32_t i, uint32_t gat) { lds_read64(a, vp); lds_read64(g, gat + (i << RSH)); vp = (vp + 8u < vlast) ? vp + 8u : vlast; }
};
__device__ __forceinline__ void wait2(uint32_t& i0, uint32_t& i1, real& a0, real& g0, real& a1, real& g1) {
  asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(i0), "+v"(i1), "+v"(a0), "+v"(g0), "+v"(a1), "+v"(g1));
}
__device__ __forceinline__ void wait2(uint32_t& i0, uint32_t& i1) { asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(i0), "+v"(i1)); }
__device__ __forceinline__ void wait0_4(real& a0, real& g0, real& a1, real& g1) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(g0), "+v"(a1), "+v"(g1)); }

__device__ __forceinline__ void rowA_x2(uint32_t i0, uint32_t v0, int len0, uint32_t i1, uint32_t v1, int len1, int L0, int L1, uint32_t gat, real& out0, real& out1) {
  real s0 = 0.0, s1 = 0.0;
  if (L0 <= 0) { out0 = 0.0; out1 = 0.0; return; }
  StreamA S0, S1;
  S0.init(i0, v0, len0);
  if (len1 > 0) S1.init(i1, v1, len1); else S1.init(i0, v0, len0);     // a lane without a second row re-reads its first (valid addresses, nothing added)
  if (len0 <= 0) S0.init(i1, v1, len1);
  uint32_t iA0 = 0, iB0 = 0, iA1 = 0, iB1 = 0;
  real aA0 = 0, gA0 = 0, aB0 = 0, gB0 = 0, aA1 = 0, gA1 = 0, aB1 = 0, gB1 = 0;
  int k = 0;
  if (L1 > 0) {
    S0.load_idx(iA0); S1.load_idx(iA1);
    S0.load_idx(iB0); S1.load_idx(iB1);
    wait2(iA0, iA1);
    S0.load_ag(aA0, gA0, iA0, gat); S1.load_ag(aA1, gA1, iA1, gat);
    for (;;) {
      S0.load_idx(iA0); S1.load_idx(iA1);
      wait2(iB0, iB1, aA0, gA0, aA1, gA1);
      S0.load_ag(aB0, gB0, iB0, gat); S1.load_ag(aB1, gB1, iB1, gat);
      asm volatile("" : "+v"(aA0), "+v"(gA0), "+v"(aA1), "+v"(gA1));
      if (k < len0) s0 += aA0 * gA0;
      if (k < len1) s1 += aA1 * gA1;
      if (++k >= L1) {                 // uniform: stream 1 is done; its pending loads are surplus.  Stream 0: G(k) in B, P(k+1) in A
        if (k >= L0) { wait0_4(aB0, gB0, aB1, gB1); goto done; }
        // hand over to the single-stream loop in the state "G(k) in B, P(k+1) in A": swap the names by running its second half first
        for (;;) {
          S0.load_idx(iB0);
          lds_wait1(iA0, aB0, gB0);
          S0.load_ag(aA0, gA0, iA0, gat);
          asm volatile("" : "+v"(aB0), "+v"(gB0));
          if (k < len0) s0 += aB0 * gB0;
          if (++k >= L0) break;
          S0.load_idx(iA0);
          lds_wait1(iB0, aA0, gA0);
          S0.load_ag(aB0, gB0, iB0, gat);
          asm volatile("" : "+v"(aA0), "+v"(gA0));
          if (k < len0) s0 += aA0 * gA0;
          if (++k >= L0) break;
        }
        wait0_4(aA0, gA0, aB0, gB0);
        goto done;
      }
      S0.load_idx(iB0); S1.load_idx(iB1);
      wait2(iA0, iA1, aB0, gB0, aB1, gB1);
      S0.load_ag(aA0, gA0, iA0, gat); S1.load_ag(aA1, gA1, iA1, gat);
      asm volatile("" : "+v"(aB0), "+v"(gB0), "+v"(aB1), "+v"(gB1));
      if (k < len0) s0 += aB0 * gB0;
      if (k < len1) s1 += aB1 * gB1;
      if (++k >= L1) {                 // state: G(k) in A, P(k+1) in B
        if (k >= L0) { wait0_4(aA0, gA0, aA1, gA1); goto done; }
        for (;;) {
          S0.load_idx(iA0);
          lds_wait1(iB0, aA0, gA0);
          S0.load_ag(aB0, gB0, iB0, gat);
          asm volatile("" : "+v"(aA0), "+v"(gA0));
          if (k < len0) s0 += aA0 * gA0;
          if (++k >= L0) break;
          S0.load_idx(iB0);
          lds_wait1(iA0, aB0, gB0);
          S0.load_ag(aA0, gA0, iA0, gat);
          asm volatile("" : "+v"(aB0), "+v"(gB0));
          if (k < len0) s0 += aB0 * gB0;
          if (++k >= L0) break;
        }
        wait0_4(aA0, gA0, aB0, gB0);
        goto done;
      }
    }
  } else {
    // no second rows in this wave: the single-stream loop with a uniform trip count
    S0.load_idx(iA0);
    S0.load_idx(iB0);
    lds_wait1(iA0);
    S0.load_ag(aA0, gA0, iA0, gat);
    for (;;) {
      S0.load_idx(iA0);
      lds_wait1(iB0, aA0, gA0);
      S0.load_ag(aB0, gB0, iB0, gat);
      asm volatile("" : "+v"(aA0), "+v"(gA0));
      if (k < len0) s0 += aA0 * gA0;
      if (++k >= L0) break;
      S0.load_idx(iB0);
      lds_wait1(iA0, aB0, gB0);
      S0.load_ag(aA0, gA0, iA0, gat);
      asm volatile("" : "+v"(aB0), "+v"(gB0));
      if (k < len0) s0 += aB0 * gB0;
      if (++k >= L0) break;
    }
    wait0_4(aA0, gA0, aB0, gB0);
  }
done:
  out0 = s0; out1 = s1;
}

// ---- A' row, NZ nonzeros per trip (NZ = 2: one ds_read2_b32 for the index pair; NZ = 4: two) ---------------------------------------------------
// K = uniform number of trips = ceil(wave max len / NZ); the index array is padded by NZ entries behind its end
template <int NZ>
__device__ __forceinline__ real rowT_multi(uint32_t idx, const uint32_t val, const uint32_t gat, const int len, const int K) {
  if (K <= 0) return 0.0;
  const int lp = len > 0 ? (len - 1) / NZ : 0;                     // last trip that holds a nonzero of this lane
  const uint32_t last = idx + 4u * NZ * (uint32_t)lp;
  real s = 0.0;
  // the index pairs stay 64-bit registers until their wait: a split right behind the load would be a register copy of data still in flight
  uint64_t iA[NZ / 2], iB[NZ / 2];
  real aA[NZ], gA[NZ], aB[NZ], gB[NZ];
#pragma unroll
  for (int t = 0; t < NZ / 2; ++t) iA[t] = iB[t] = 0;
#pragma unroll
  for (int t = 0; t < NZ; ++t) aA[t] = gA[t] = aB[t] = gB[t] = 0.0;
  auto load_idx = [&](uint64_t* d) {
#pragma unroll
    for (int t = 0; t < NZ / 2; ++t) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d[t]) : "v"(idx + 8u * t));
    idx = (idx + 4u * NZ < last) ? idx + 4u * NZ : last;
  };
  auto load_ag = [&](real* a, real* g, const uint64_t* i) {
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
      const uint32_t e = (t & 1) ? (uint32_t)(i[t / 2] >> 32) : (uint32_t)i[t / 2];
      lds_read64(a[t], val + ((e & 0xffffu) << RSH)); lds_read64(g[t], gat + ((e >> 16) << RSH));
    }
  };
  // the waits: after issuing the NZ/2 index instructions of trip k+2, everything older must have arrived
  auto wait_idx = [&](uint64_t* i, real* a, real* g) {
    if (NZ == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(i[0]), "+v"(a[0]), "+v"(g[0]), "+v"(a[1]), "+v"(g[1]));
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(i[0]), "+v"(i[NZ / 2 - 1]), "+v"(a[0]), "+v"(g[0]), "+v"(a[1]), "+v"(g[1]), "+v"(a[NZ - 2]), "+v"(g[NZ - 2]), "+v"(a[NZ - 1]), "+v"(g[NZ - 1]));
  };
  auto tie = [&](real* a, real* g) {
#pragma unroll
    for (int t = 0; t < NZ; ++t) asm volatile("" : "+v"(a[t]), "+v"(g[t]));
  };
  load_idx(iA);
  load_idx(iB);
  if (NZ == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(iA[0])); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(iA[0]), "+v"(iA[NZ / 2 - 1]));
  load_ag(aA, gA, iA);
  int k = 0;
  for (;;) {
    load_idx(iA);
    wait_idx(iB, aA, gA);
    load_ag(aB, gB, iB);
    tie(aA, gA);
#pragma unroll
    for (int t = 0; t < NZ; ++t) if (NZ * k + t < len) s += aA[t] * gA[t];
    if (++k >= K) break;
    load_idx(iB);
    wait_idx(iA, aB, gB);
    load_ag(aA, gA, iA);
    tie(aB, gB);
#pragma unroll
    for (int t = 0; t < NZ; ++t) if (NZ * k + t < len) s += aB[t] * gB[t];
    if (++k >= K) break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  tie(aA, gA); tie(aB, gB);
  return s;
}

// ---- lean forms: fewer instructions per nonzero -----------------------------------------------------------------------------------------------
// A rows: the u16 entries are BYTE OFFSETS of the gathered operand (position * 8; xv sits at LDS address 0, so the loaded index IS the address),
// value / index pointers advance by immediate offsets (one add per pointer and two nonzeros, no clamp: the arrays are padded), the trip count is the
// wave's longest row (scalar loop control), a product past the lane's row end is not added.
#define DSR(op, dst, addr, off) asm volatile(op " %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
__device__ __forceinline__ real rowA_lean(uint32_t ip, uint32_t vp, const int len, const int L) {
  if (L <= 0) return 0.0;
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSR("ds_read_u16", iA, ip, 0);
  DSR("ds_read_u16", iB, ip, 2);
  lds_wait1(iA);
  DSR("ds_read_b64", aA, vp, 0);
  DSR("ds_read_b64", gA, iA, 0);
  int k = 0;
  for (;;) {
    DSR("ds_read_u16", iA, ip, 4);
    lds_wait1(iB, aA, gA);
    DSR("ds_read_b64", aB, vp, 8);
    DSR("ds_read_b64", gB, iB, 0);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    DSR("ds_read_u16", iB, ip, 6);
    lds_wait1(iA, aB, gB);
    DSR("ds_read_b64", aA, vp, 16);
    DSR("ds_read_b64", gA, iA, 0);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += 4u; vp += 16u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
// A' rows: packed entry = (LDS byte address of the value, 18 bits) | (row position << 21): value address = e & 0x3ffff, gather address = (e >> 18) + TV_OFF
#define TV_OFF 4096
#define DSR_TV(dst, addr) asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(dst) : "v"(addr))
__device__ __forceinline__ real rowT_lean(uint32_t ip, const int len, const int L) {
  if (L <= 0) return 0.0;
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSR("ds_read_b32", iA, ip, 0);
  DSR("ds_read_b32", iB, ip, 4);
  lds_wait1(iA);
  DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
  DSR_TV(gA, iA >> 18);
  int k = 0;
  for (;;) {
    DSR("ds_read_b32", iA, ip, 8);
    lds_wait1(iB, aA, gA);
    DSR("ds_read_b64", aB, iB & 0x3ffffu, 0);
    DSR_TV(gB, iB >> 18);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    DSR("ds_read_b32", iB, ip, 12);
    lds_wait1(iA, aB, gB);
    DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
    DSR_TV(gA, iA >> 18);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += 8u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
// A' rows, lean, two nonzeros per trip (K = ceil(L / 2) trips)
__device__ __forceinline__ real rowT_lean2(uint32_t ip, const int len, const int K) {
  if (K <= 0) return 0.0;
  real s = 0.0;
  uint64_t iA = 0, iB = 0;
  real aA0 = 0, gA0 = 0, aA1 = 0, gA1 = 0, aB0 = 0, gB0 = 0, aB1 = 0, gB1 = 0;
  auto ld = [&](real& a0, real& g0, real& a1, real& g1, uint64_t i) {
    const uint32_t e0 = (uint32_t)i, e1 = (uint32_t)(i >> 32);
    DSR("ds_read_b64", a0, e0 & 0x3ffffu, 0); DSR_TV(g0, e0 >> 18);
    DSR("ds_read_b64", a1, e1 & 0x3ffffu, 0); DSR_TV(g1, e1 >> 18);
  };
  asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(iA) : "v"(ip));
  asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(iB) : "v"(ip));
  asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(iA));
  ld(aA0, gA0, aA1, gA1, iA);
  int k = 0;
  for (;;) {
    asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:5" : "=v"(iA) : "v"(ip));
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(iB), "+v"(aA0), "+v"(gA0), "+v"(aA1), "+v"(gA1));
    ld(aB0, gB0, aB1, gB1, iB);
    asm volatile("" : "+v"(aA0), "+v"(gA0), "+v"(aA1), "+v"(gA1));
    if (2 * k < len) s += aA0 * gA0;
    if (2 * k + 1 < len) s += aA1 * gA1;
    if (++k >= K) break;
    asm volatile("ds_read2_b32 %0, %1 offset0:6 offset1:7" : "=v"(iB) : "v"(ip));
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(iA), "+v"(aB0), "+v"(gB0), "+v"(aB1), "+v"(gB1));
    ld(aA0, gA0, aA1, gA1, iA);
    asm volatile("" : "+v"(aB0), "+v"(gB0), "+v"(aB1), "+v"(gB1));
    ip += 16u;
    if (2 * k < len) s += aB0 * gB0;
    if (2 * k + 1 < len) s += aB1 * gB1;
    if (++k >= K) break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aA0), "+v"(gA0), "+v"(aA1), "+v"(gA1));
  asm volatile("" : "+v"(aB0), "+v"(gB0), "+v"(aB1), "+v"(gB1));
  return s;
}

// ---- lean + SLICED layout: nonzero t of the 64 rows a wave works on in one trip are NEIGHBOURS (entry (t, lane) of a wave-slot at t * 64 + lane):
// every value / index read of a trip is one linear wave access, whatever the row lengths (rows sorted by length make the row-major stride of a wave
// EQUAL to the row length: lengths 8, 12, 16 put 4 ... 16 lanes on one bank pair)
#ifndef SL
#define SL 32                  // lanes per slice: 64 = one slice per wave-slot, 32 = one per half-wave (half the padding; a b64 wave read is served 32 lanes at a time)
#endif
#define XSTR(x) #x
#define DSRO(op, dst, addr, off) asm volatile(op " %0, %1 offset:" XSTR(off) : "=v"(dst) : "v"(addr))
__device__ __forceinline__ real rowA_sliced(uint32_t ip, uint32_t vp, const int len, const int L) {
  if (L <= 0) return 0.0;
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSRO("ds_read_u16", iA, ip, 0);
  DSRO("ds_read_u16", iB, ip, (SL * 2));
  lds_wait1(iA);
  DSRO("ds_read_b64", aA, vp, 0);
  DSRO("ds_read_b64", gA, iA, 0);
  int k = 0;
  for (;;) {
    DSRO("ds_read_u16", iA, ip, (SL * 4));
    lds_wait1(iB, aA, gA);
    DSRO("ds_read_b64", aB, vp, (SL * 8));
    DSRO("ds_read_b64", gB, iB, 0);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    DSRO("ds_read_u16", iB, ip, (SL * 6));
    lds_wait1(iA, aB, gB);
    DSRO("ds_read_b64", aA, vp, (SL * 16));
    DSRO("ds_read_b64", gA, iA, 0);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += SL * 4u; vp += SL * 16u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
__device__ __forceinline__ real rowT_sliced(uint32_t ip, const int len, const int L) {
  if (L <= 0) return 0.0;
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSRO("ds_read_b32", iA, ip, 0);
  DSRO("ds_read_b32", iB, ip, (SL * 4));
  lds_wait1(iA);
  DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
  DSR_TV(gA, iA >> 18);
  int k = 0;
  for (;;) {
    DSRO("ds_read_b32", iA, ip, (SL * 8));
    lds_wait1(iB, aA, gA);
    DSR("ds_read_b64", aB, iB & 0x3ffffu, 0);
    DSR_TV(gB, iB >> 18);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    DSRO("ds_read_b32", iB, ip, (SL * 12));
    lds_wait1(iA, aB, gB);
    DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
    DSR_TV(gA, iA >> 18);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += SL * 8u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}

// ---- lean + JAGGED layout (JDS): nonzero t of the rows of a wave-slot that HAVE a nonzero t are neighbours, no padding.  The rows of a slot are
// sorted by decreasing length, so the lanes still active in trip t are a prefix [0, cnt_t) of the wave and entry (t, lane) sits at E(t) + lane with
// E(t + 1) = E(t) + cnt_t; cnt_t = popcount(ballot(t < len)) is a scalar the loop computes as it goes (one v_cmp + s_bcnt1 per trip).
// Every value / index read of a trip is one linear wave access; an idle lane reads the entries behind the prefix (valid, unused).
__device__ __forceinline__ real rowA_jds(const uint32_t lane_val, const uint32_t lane_idx, const int E0, const int Emax, const int len, const int L) {
  if (L <= 0) return 0.0;
  auto cnt = [&](int t) -> int { return (int)__builtin_popcountll(__ballot(t < len)); };
  int e0 = E0, e1 = min(e0 + cnt(0), Emax), e2 = min(e1 + cnt(1), Emax);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSR("ds_read_u16", iA, lane_idx + 2u * (uint32_t)e0, 0);
  DSR("ds_read_u16", iB, lane_idx + 2u * (uint32_t)e1, 0);
  lds_wait1(iA);
  DSR("ds_read_b64", aA, lane_val + 8u * (uint32_t)e0, 0);
  DSR("ds_read_b64", gA, iA, 0);
  int k = 0;
  for (;;) {
    DSR("ds_read_u16", iA, lane_idx + 2u * (uint32_t)e2, 0);
    lds_wait1(iB, aA, gA);
    DSR("ds_read_b64", aB, lane_val + 8u * (uint32_t)e1, 0);
    DSR("ds_read_b64", gB, iB, 0);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    e1 = e2; e2 = min(e2 + cnt(k + 2), Emax);
    if (++k >= L) break;
    DSR("ds_read_u16", iB, lane_idx + 2u * (uint32_t)e2, 0);
    lds_wait1(iA, aB, gB);
    DSR("ds_read_b64", aA, lane_val + 8u * (uint32_t)e1, 0);
    DSR("ds_read_b64", gA, iA, 0);
    asm volatile("" : "+v"(aB), "+v"(gB));
    if (k < len) s += aB * gB;
    e1 = e2; e2 = min(e2 + cnt(k + 2), Emax);
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
__device__ __forceinline__ real rowT_jds(const uint32_t lane_idx, const int E0, const int Emax, const int len, const int L) {
  if (L <= 0) return 0.0;
  auto cnt = [&](int t) -> int { return (int)__builtin_popcountll(__ballot(t < len)); };
  int e0 = E0, e1 = min(e0 + cnt(0), Emax), e2 = min(e1 + cnt(1), Emax);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  DSR("ds_read_b32", iA, lane_idx + 4u * (uint32_t)e0, 0);
  DSR("ds_read_b32", iB, lane_idx + 4u * (uint32_t)e1, 0);
  lds_wait1(iA);
  DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
  DSR_TV(gA, iA >> 18);
  int k = 0;
  for (;;) {
    DSR("ds_read_b32", iA, lane_idx + 4u * (uint32_t)e2, 0);
    lds_wait1(iB, aA, gA);
    DSR("ds_read_b64", aB, iB & 0x3ffffu, 0);
    DSR_TV(gB, iB >> 18);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    e2 = min(e2 + cnt(k + 2), Emax);
    if (++k >= L) break;
    DSR("ds_read_b32", iB, lane_idx + 4u * (uint32_t)e2, 0);
    lds_wait1(iA, aB, gB);
    DSR("ds_read_b64", aA, iA & 0x3ffffu, 0);
    DSR_TV(gA, iA >> 18);
    asm volatile("" : "+v"(aB), "+v"(gB));
    if (k < len) s += aB * gB;
    e2 = min(e2 + cnt(k + 2), Emax);
    if (++k >= L) break;
  }
  lds_wait0(aA, gA); asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

struct LabIn {
  const real* Aval; const unsigned short* Acol; const uint32_t* Tpr;   // stored-sorted arrays (global memory), nnz + pad entries
  const int *a0, *a1;      // [2][2][BS]: pairing p, slot j: bounds of the row thread t computes (a0 == a1: none)
  const int* arow;         // [2][2][BS]: its original row (-1: none)
  const int *t0, *t1;      // [BS]: bounds of the column thread t computes
  const int* tcol;         // [BS]: its original column (-1: none)
  int n, m, nnz, pad;
  // sliced layout (A pass form 4, A' pass form 5)
  const real* AvalS; const unsigned short* AcolS; const uint32_t* TprS; const int *sbA, *sbT; int nS, nTS;
  // jagged layout (A pass form 6, A' pass form 7): values / byte offsets / pairs in JDS order, first entry of every thread's wave-slot, rows of a thread, lengths
  const real* AvalJ; const unsigned short* AcolJ; const uint32_t* TprJ; const int *jE, *jlen, *jrow, *jET;   // entries of the sliced value / index arrays, of the sliced pair array
};

template <int VA, int VT>
__global__ __launch_bounds__(BS) void k_lab(LabIn in, int reps, double* out, unsigned long long* cyc) {
  extern __shared__ real lds[];
  const int tid = threadIdx.x;
  real* xv = lds;                      // n  (<= 512)
  real* tv = lds + 512;                // m  (<= 1024)
  real* Aval = lds + 512 + 1024;
  unsigned short* Acol = reinterpret_cast<unsigned short*>(Aval + in.nnz + in.pad);
  uint32_t* Tpr = reinterpret_cast<uint32_t*>(Acol + ((in.nnz + in.pad + 3) / 4) * 4);
  constexpr bool LEAN_A = (VA == 3), LEAN_T = (VT >= 3);
  constexpr bool SLICED = (VA == 4 || VT == 5);
  constexpr bool JDS = (VA == 6 || VT == 7);
  if constexpr (JDS) {
    for (int i = tid; i < in.nnz + in.pad; i += BS) {
      Aval[i] = in.AvalJ[i]; Acol[i] = in.AcolJ[i];
      const uint32_t e = in.TprJ[i]; Tpr[i] = (lds_addr_of(Aval) + ((e & 0xffffu) << 3)) | ((e >> 16) << 21);
    }
    if (lds_addr_of(lds) != 0u || lds_addr_of(Aval) + 8u * (uint32_t)(in.nnz + in.pad) >= (1u << 18)) { if (tid == 0) cyc[2] = 1; return; }
  } else
  if constexpr (SLICED) {
    // sliced arrays replace the row-major ones (both passes must use them: the values move)
    Acol = reinterpret_cast<unsigned short*>(Aval + in.nS);
    Tpr = reinterpret_cast<uint32_t*>(Acol + ((in.nS + 3) / 4) * 4);
    for (int i = tid; i < in.nS; i += BS) { Aval[i] = in.AvalS[i]; Acol[i] = in.AcolS[i]; }
    for (int i = tid; i < in.nTS; i += BS) { const uint32_t e = in.TprS[i]; Tpr[i] = (lds_addr_of(Aval) + ((e & 0xffffu) << 3)) | ((e >> 16) << 21); }
    if (lds_addr_of(lds) != 0u || lds_addr_of(Aval) + 8u * (uint32_t)in.nS >= (1u << 18)) { if (tid == 0) cyc[2] = 1; return; }
  } else
  for (int i = tid; i < in.nnz + in.pad; i += BS) {
    Aval[i] = in.Aval[i];
    Acol[i] = LEAN_A ? (unsigned short)(in.Acol[i] * 8) : in.Acol[i];                      // lean: byte offset of the gathered operand
    const uint32_t e = in.Tpr[i];
    Tpr[i] = LEAN_T ? ((lds_addr_of(Aval) + ((e & 0xffffu) << 3)) | ((e >> 16) << 21)) : e;   // lean: value address | row << 21
  }
  if ((LEAN_A || LEAN_T) && (lds_addr_of(lds) != 0u || lds_addr_of(Aval) + 8u * (uint32_t)(in.nnz + in.pad) >= (1u << 18))) { if (tid == 0) cyc[2] = 1; return; }
  for (int i = tid; i < 512; i += BS) xv[i] = 1.0 + 1e-3 * i;
  for (int i = tid; i < 1024; i += BS) tv[i] = 0.0;
  constexpr int PAIRING = (VA == 2) ? 1 : 0;
  int ka0[2], ka1[2], ra[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int o = (PAIRING * 2 + j) * BS + tid; ka0[j] = in.a0[o]; ka1[j] = in.a1[o]; ra[j] = in.arow[o]; }
  const int kt0 = in.t0[tid], kt1 = in.t1[tid], ct = in.tcol[tid];
  const int LA0 = wave_max_i(max(ka1[0] - ka0[0], ka1[1] - ka0[1])), LA1 = wave_max_i(ka1[1] - ka0[1]);
  const int LT = wave_max_i(kt1 - kt0);
  const int jE[2] = {JDS ? in.jE[tid] : 0, JDS ? in.jE[BS + tid] : 0}, jlen[2] = {JDS ? in.jlen[tid] : 0, JDS ? in.jlen[BS + tid] : 0};
  const int jrow[2] = {JDS ? in.jrow[tid] : -1, JDS ? in.jrow[BS + tid] : -1}, jET = JDS ? in.jET[tid] : 0;
  const int LJ[2] = {wave_max_i(jlen[0]), wave_max_i(jlen[1])};
  const int lane = tid & 63;
  const int sbA[2] = {SLICED ? in.sbA[tid] : 0, SLICED ? in.sbA[BS + tid] : 0}, sbT = SLICED ? in.sbT[tid] : 0;
  const int LAj[2] = {wave_max_i(ka1[0] - ka0[0]), wave_max_i(ka1[1] - ka0[1])};
  const uint32_t lA_val = lds_addr_of(Aval), lA_col = lds_addr_of(Acol), lT_pr = lds_addr_of(Tpr), l_xv = lds_addr_of(xv), l_tv = lds_addr_of(tv);
  __syncthreads();
  unsigned long long cA = 0, cT = 0;
  real u = (ct >= 0) ? xv[ct] : 0.0;
  for (int r = 0; r < reps; ++r) {
    long long c0 = clock64();
    real tmp[2];
    if (VA == 6) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        tmp[j] = rowA_jds(lds_addr_of(Aval) + 8u * (uint32_t)lane, lds_addr_of(Acol) + 2u * (uint32_t)lane, __builtin_amdgcn_readfirstlane(jE[j]), in.nnz, jlen[j], LJ[j]) + 0.0;
    } else if (VA == 4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = rowA_sliced(lds_addr_of(Acol) + 2u * (uint32_t)sbA[j], lds_addr_of(Aval) + 8u * (uint32_t)sbA[j], ka1[j] - ka0[j], LAj[j]) + 0.0;
    } else if (VA == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = rowA_lean(lA_col + 2u * (uint32_t)ka0[j], lA_val + ((uint32_t)ka0[j] << RSH), ka1[j] - ka0[j], LAj[j]) + 0.0;
    } else if (VA == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = row_pipe3<false>(lA_col + 2u * (uint32_t)ka0[j], lA_val + ((uint32_t)ka0[j] << RSH), l_xv, ka1[j] - ka0[j]) + 0.0;
    } else {
      rowA_x2(lA_col + 2u * (uint32_t)ka0[0], lA_val + ((uint32_t)ka0[0] << RSH), ka1[0] - ka0[0],
              lA_col + 2u * (uint32_t)ka0[1], lA_val + ((uint32_t)ka0[1] << RSH), ka1[1] - ka0[1], LA0, LA1, l_xv, tmp[0], tmp[1]);
      tmp[0] += 0.0; tmp[1] += 0.0;
    }
    if (VA == 6) {
#pragma unroll
      for (int j = 0; j < 2; ++j) if (jrow[j] >= 0) tv[jrow[j]] = tmp[j] * 0.1;
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) if (ra[j] >= 0) tv[ra[j]] = tmp[j] * 0.1;
    __syncthreads();
    long long c1 = clock64();
    real c;
    if (VT == 0) c = row_pipe3<true>(lT_pr + 4u * (uint32_t)kt0, lA_val, l_tv, kt1 - kt0);
    else if (VT == 1) c = rowT_multi<2>(lT_pr + 4u * (uint32_t)kt0, lA_val, l_tv, kt1 - kt0, (LT + 1) / 2);
    else if (VT == 2) c = rowT_multi<4>(lT_pr + 4u * (uint32_t)kt0, lA_val, l_tv, kt1 - kt0, (LT + 3) / 4);
    else if (VT == 7) c = rowT_jds(lds_addr_of(Tpr) + 4u * (uint32_t)lane, __builtin_amdgcn_readfirstlane(jET), in.nnz, kt1 - kt0, LT);
    else if (VT == 5) c = rowT_sliced(lds_addr_of(Tpr) + 4u * (uint32_t)sbT, kt1 - kt0, LT);
    else if (VT == 3) c = rowT_lean(lT_pr + 4u * (uint32_t)kt0, kt1 - kt0, LT);
    else c = rowT_lean2(lT_pr + 4u * (uint32_t)kt0, kt1 - kt0, (LT + 1) / 2);
    u = 0.5 * u + 1e-3 * c;
    if (ct >= 0) xv[ct] = u;
    __syncthreads();
    long long c2 = clock64();
    cA += (unsigned long long)(c1 - c0); cT += (unsigned long long)(c2 - c1);
  }
  out[tid] = u;
  if (tid == 0) { cyc[0] = cA; cyc[1] = cT; }
}

int main() {
  const int n = 500, m = 1000, nnz = 9000, pad = 64, reps = 400;   // (config 3 has 10000 nonzeros: its sliced arrays do not fit next to the rest of the image)
  srand(12345);
  std::vector<std::vector<std::pair<int, double>>> rows((size_t)m);
  for (int e = 0; e < nnz; ++e) { const int i = rand() % m, j = rand() % n; rows[(size_t)i].push_back({j, (rand() % 2001 - 1000) / 1000.0}); }
  std::vector<int> ord((size_t)m); std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return rows[(size_t)x].size() > rows[(size_t)y].size(); });
  std::vector<real> Aval((size_t)nnz + pad, 0.0); std::vector<unsigned short> Acol((size_t)nnz + pad, 0); std::vector<int> Arp((size_t)m + 1), newstart((size_t)m);
  int w = 0;
  for (int q = 0; q < m; ++q) { const int r = ord[(size_t)q]; Arp[(size_t)q] = w; newstart[(size_t)r] = w; for (auto& e : rows[(size_t)r]) { Aval[(size_t)w] = e.second; Acol[(size_t)w] = (unsigned short)e.first; ++w; } }
  Arp[(size_t)m] = w;
  // columns: (value position, row) in row order
  std::vector<std::vector<uint32_t>> cols((size_t)n);
  for (int r = 0; r < m; ++r) for (size_t t = 0; t < rows[(size_t)r].size(); ++t) cols[(size_t)rows[(size_t)r][t].first].push_back((uint32_t)(newstart[(size_t)r] + (int)t) | ((uint32_t)r << 16));
  std::vector<int> ordc((size_t)n); std::iota(ordc.begin(), ordc.end(), 0);
  std::stable_sort(ordc.begin(), ordc.end(), [&](int x, int y) { return cols[(size_t)x].size() > cols[(size_t)y].size(); });
  std::vector<uint32_t> Tpr((size_t)nnz + pad, 0u); std::vector<int> Trp((size_t)n + 1);
  w = 0;
  for (int q = 0; q < n; ++q) { Trp[(size_t)q] = w; for (uint32_t e : cols[(size_t)ordc[(size_t)q]]) Tpr[(size_t)w++] = e; }
  Trp[(size_t)n] = w;
  std::vector<int> a0(4 * BS, 0), a1(4 * BS, 0), arow(4 * BS, -1), t0(BS, 0), t1(BS, 0), tcol(BS, -1);
  for (int t = 0; t < BS; ++t) {
    const int q[2][2] = {{t, 1023 - t}, {2 * t, 2 * t + 1}};
    for (int p = 0; p < 2; ++p) for (int j = 0; j < 2; ++j) if (q[p][j] < m) { const int o = (p * 2 + j) * BS + t; a0[(size_t)o] = Arp[(size_t)q[p][j]]; a1[(size_t)o] = Arp[(size_t)q[p][j] + 1]; arow[(size_t)o] = ord[(size_t)q[p][j]]; }
    if (t < n) { t0[(size_t)t] = Trp[(size_t)t]; t1[(size_t)t] = Trp[(size_t)t + 1]; tcol[(size_t)t] = ordc[(size_t)t]; }
  }
  printf("config-3 sized problem: m = %d rows (longest %zu, shortest %zu nonzeros), n = %d columns (longest %zu, shortest %zu), %d nonzeros\n", m, rows[(size_t)ord[0]].size(),
         rows[(size_t)ord[(size_t)m - 1]].size(), n, cols[(size_t)ordc[0]].size(), cols[(size_t)ordc[(size_t)n - 1]].size(), nnz);
  // sliced layout: slice = SL consecutive threads of one slot (pairing t / 1023 - t); entry (trip, lane in slice) at off + trip * SL + lane
  std::vector<real> AvalS; std::vector<unsigned short> AcolS; std::vector<int> sbA(2 * BS, 0), sbT(BS, 0), newpos((size_t)nnz, 0);
  for (int j = 0; j < 2; ++j) for (int sl = 0; sl < BS / SL; ++sl) {
    int L = 0;
    for (int l = 0; l < SL; ++l) { const int o = j * BS + SL * sl + l; L = std::max(L, a1[(size_t)o] - a0[(size_t)o]); }
    const int off = (int)AvalS.size();
    AvalS.resize((size_t)off + SL * (size_t)L, 0.0); AcolS.resize(AvalS.size(), 0);
    for (int l = 0; l < SL; ++l) {
      const int o = j * BS + SL * sl + l; sbA[(size_t)o] = off + l;
      for (int t = a0[(size_t)o]; t < a1[(size_t)o]; ++t) { const int e = off + (t - a0[(size_t)o]) * SL + l; AvalS[(size_t)e] = Aval[(size_t)t]; AcolS[(size_t)e] = (unsigned short)(Acol[(size_t)t] * 8); newpos[(size_t)t] = e; }
    }
  }
  AvalS.resize(AvalS.size() + 40 * SL, 0.0); AcolS.resize(AvalS.size(), 0);      // the loops read ahead and run to the wave's longest row: into the next slices, or into this tail
  std::vector<uint32_t> TprS;
  for (int sl = 0; sl < BS / SL; ++sl) {
    int L = 0;
    for (int l = 0; l < SL; ++l) L = std::max(L, t1[(size_t)(SL * sl + l)] - t0[(size_t)(SL * sl + l)]);
    const int off = (int)TprS.size();
    TprS.resize((size_t)off + SL * (size_t)L, 0u);
    for (int l = 0; l < SL; ++l) {
      const int t = SL * sl + l; sbT[(size_t)t] = off + l;
      for (int e = t0[(size_t)t]; e < t1[(size_t)t]; ++e) TprS[(size_t)off + (size_t)(e - t0[(size_t)t]) * SL + l] = (uint32_t)newpos[(size_t)(Tpr[(size_t)e] & 0xffffu)] | (Tpr[(size_t)e] & 0xffff0000u);
    }
  }
  const size_t slicedA = AvalS.size() - 40 * SL, slicedT = TprS.size();
  TprS.resize(TprS.size() + 40 * SL, 0u);
  printf("sliced layout, %d lanes per slice: %zu value slots (+%.1f %%), %zu pair slots (+%.1f %%) for %d nonzeros\n", SL, slicedA, 100.0 * ((double)slicedA / nnz - 1.0), slicedT, 100.0 * ((double)slicedT / nnz - 1.0), nnz);
  // jagged layout: slot 0 of thread t = sorted position t, slot 1 = 512 + 64 (7 - wave) + lane (both descending inside a wave, the waves of slot 1 in reverse order)
  std::vector<real> AvalJ((size_t)nnz + pad, 0.0); std::vector<unsigned short> AcolJ((size_t)nnz + pad, 0); std::vector<uint32_t> TprJ((size_t)nnz + pad, 0u);
  std::vector<int> jE(2 * BS, 0), jlen(2 * BS, 0), jrow(2 * BS, -1), jET(BS, 0), newposJ((size_t)nnz, 0);
  {
    int E = 0;
    for (int j = 0; j < 2; ++j) for (int wv = 0; wv < 8; ++wv) {
      int q[64], ln[64], L = 0;
      for (int l = 0; l < 64; ++l) { q[l] = j == 0 ? 64 * wv + l : 512 + 64 * (7 - wv) + l; ln[l] = q[l] < m ? Arp[(size_t)q[l] + 1] - Arp[(size_t)q[l]] : 0; L = std::max(L, ln[l]); }
      for (int l = 0; l < 64; ++l) { const int o = j * BS + 64 * wv + l; jE[(size_t)o] = E; jlen[(size_t)o] = ln[l]; jrow[(size_t)o] = q[l] < m ? ord[(size_t)q[l]] : -1; }
      for (int t = 0; t < L; ++t) for (int l = 0; l < 64; ++l) if (ln[l] > t) {
        const int src = Arp[(size_t)q[l]] + t;
        AvalJ[(size_t)E] = Aval[(size_t)src]; AcolJ[(size_t)E] = (unsigned short)(Acol[(size_t)src] * 8); newposJ[(size_t)src] = E; ++E;
      }
    }
    E = 0;
    for (int wv = 0; wv < 8; ++wv) {
      int L = 0;
      for (int l = 0; l < 64; ++l) L = std::max(L, t1[(size_t)(64 * wv + l)] - t0[(size_t)(64 * wv + l)]);
      for (int l = 0; l < 64; ++l) jET[(size_t)(64 * wv + l)] = E;
      for (int t = 0; t < L; ++t) for (int l = 0; l < 64; ++l) { const int c = 64 * wv + l; if (t1[(size_t)c] - t0[(size_t)c] > t) { const uint32_t e = Tpr[(size_t)(t0[(size_t)c] + t)]; TprJ[(size_t)E++] = (uint32_t)newposJ[(size_t)(e & 0xffffu)] | (e & 0xffff0000u); } }
    }
  }
  LabIn in;
  real* dAval; unsigned short* dAcol; uint32_t* dTpr; int *da0, *da1, *darow, *dt0, *dt1, *dtcol; double* dout; unsigned long long* dcyc;
#define UP(dst, v) (void)hipMalloc((void**)&dst, (v).size() * sizeof((v)[0])); (void)hipMemcpy(dst, (v).data(), (v).size() * sizeof((v)[0]), hipMemcpyHostToDevice)
  real* dAvalS; unsigned short* dAcolS; uint32_t* dTprS; int *dsbA, *dsbT;
  UP(dAvalS, AvalS); UP(dAcolS, AcolS); UP(dTprS, TprS); UP(dsbA, sbA); UP(dsbT, sbT);
  in.AvalS = dAvalS; in.AcolS = dAcolS; in.TprS = dTprS; in.sbA = dsbA; in.sbT = dsbT; in.nS = (int)AvalS.size(); in.nTS = (int)TprS.size();
  real* dAvalJ; unsigned short* dAcolJ; uint32_t* dTprJ; int *djE, *djlen, *djrow, *djET;
  UP(dAvalJ, AvalJ); UP(dAcolJ, AcolJ); UP(dTprJ, TprJ); UP(djE, jE); UP(djlen, jlen); UP(djrow, jrow); UP(djET, jET);
  in.AvalJ = dAvalJ; in.AcolJ = dAcolJ; in.TprJ = dTprJ; in.jE = djE; in.jlen = djlen; in.jrow = djrow; in.jET = djET;
  UP(dAval, Aval); UP(dAcol, Acol); UP(dTpr, Tpr); UP(da0, a0); UP(da1, a1); UP(darow, arow); UP(dt0, t0); UP(dt1, t1); UP(dtcol, tcol);
  (void)hipMalloc((void**)&dout, BS * 8); (void)hipMalloc((void**)&dcyc, 32); (void)hipMemset(dcyc, 0, 32);
  in.Aval = dAval; in.Acol = dAcol; in.Tpr = dTpr; in.a0 = da0; in.a1 = da1; in.arow = darow; in.t0 = dt0; in.t1 = dt1; in.tcol = dtcol; in.n = n; in.m = m; in.nnz = nnz; in.pad = pad;
  const size_t shm0 = (512 + 1024) * 8 + (size_t)(nnz + pad) * 8 + (size_t)((nnz + pad + 3) / 4) * 4 * 2 + (size_t)(nnz + pad) * 4;
  const size_t shmS = (512 + 1024) * 8 + AvalS.size() * 8 + ((AvalS.size() + 3) / 4) * 4 * 2 + TprS.size() * 4;
  const size_t shm = std::max(shm0, shmS);
  printf("dynamic LDS %zu bytes; %d repetitions of (A pass, barrier, A' pass, barrier); shader-clock cycles per pass\n", shm, reps);
  std::vector<double> ref;
  auto run = [&](const void* fn, const char* name) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    void* args[] = {&in, (void*)&reps, &dout, &dcyc};
    for (int rep = 0; rep < 2; ++rep) { (void)hipLaunchKernel(fn, dim3(1), dim3(BS), args, shm, 0); (void)hipDeviceSynchronize(); }
    hipError_t e = hipGetLastError();
    unsigned long long c[3]; std::vector<double> o(BS);
    (void)hipMemcpy(c, dcyc, 24, hipMemcpyDeviceToHost); if (c[2]) printf("  (layout check of the lean form failed)\n"); (void)hipMemcpy(o.data(), dout, BS * 8, hipMemcpyDeviceToHost);
    if (ref.empty()) ref = o;
    int same = 1; for (int i = 0; i < BS; ++i) if (o[(size_t)i] != ref[(size_t)i]) same = 0;
    printf("  %-62s A pass %7.0f   A' pass %7.0f   sum %7.0f   bits %s  (%s)\n", name, (double)c[0] / reps, (double)c[1] / reps, (double)(c[0] + c[1]) / reps, same ? "equal" : "DIFFER", hipGetErrorString(e));
  };
  run((const void*)k_lab<0, 0>, "A: rows one after the other        A': 1 nonzero per trip");
  run((const void*)k_lab<1, 0>, "A: two rows in lockstep (t, 1023-t) A': 1 nonzero per trip");
  run((const void*)k_lab<2, 0>, "A: two rows in lockstep (2t, 2t+1)  A': 1 nonzero per trip");
  run((const void*)k_lab<0, 1>, "A: rows one after the other        A': 2 nonzeros per trip");
  run((const void*)k_lab<0, 2>, "A: rows one after the other        A': 4 nonzeros per trip");
  run((const void*)k_lab<1, 1>, "A: two rows in lockstep (t, 1023-t) A': 2 nonzeros per trip");
  run((const void*)k_lab<1, 2>, "A: two rows in lockstep (t, 1023-t) A': 4 nonzeros per trip");
  run((const void*)k_lab<4, 5>, "A: lean + sliced layout              A': lean + sliced pairs");
  run((const void*)k_lab<6, 7>, "A: lean + jagged layout (no padding) A': lean + jagged pairs");
  run((const void*)k_lab<3, 0>, "A: lean loop, rows one after the other  A': 1 nonzero per trip");
  run((const void*)k_lab<0, 3>, "A: rows one after the other        A': lean, 1 nonzero per trip");
  run((const void*)k_lab<0, 4>, "A: rows one after the other        A': lean, 2 nonzeros per trip");
  run((const void*)k_lab<3, 3>, "A: lean                             A': lean, 1 nonzero per trip");
  run((const void*)k_lab<3, 4>, "A: lean                             A': lean, 2 nonzeros per trip");
  return 0;
}
