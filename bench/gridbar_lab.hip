// gridbar_lab.hip -- lab for VERDICT r03 items 2 / 3b: what does a software barrier cost
//   (A) among W = 2, 4, 8 workgroups of ONE XCD (the "wide" form of the batch kernel: a straggler problem's sparse passes split over a few CUs;
//       L2 of the XCD is the coherence point: plain stores + sc1 loads, no fences), and
//   (B) among ALL 256 CUs of the chip (a whole-chip persistent CG on the assembled operator of BASELINE config 5: n = 50 000, 708 k nonzeros), where
//       the eight L2s are NOT coherent with each other: every publish needs an agent-scope release (L2 write-back) and every consume an acquire
//       (L2 invalidate).  Measured per round = publish a slice of an n-vector + barrier + gather ~11 entries per thread of the whole vector from
//       the other workgroups' slices (what one half of a Krylov iteration does), flat counter vs per-XCD counters + one global counter.
// Kill criterion written down in advance (VERDICT): (B) >= 3 us per barrier rules the whole-chip persistent CG out (two barriers per Krylov
// iteration against today's 12.7 us for the two-launch iteration).
//   hipcc --offload-arch=gfx950 -O3 bench/gridbar_lab.hip -o bench/gridbar_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xF; }

// ---- (A) in-XCD groups of W workgroups: NG groups run concurrently on XCD `want`; each round = store + barrier + neighbour read + barrier
__global__ __launch_bounds__(512) void k_group_barrier(unsigned* sync, int W, int NG, int want, int rounds, double* data, unsigned long long* out, int* errs) {
  __shared__ int s_rank, s_fail;
  if (threadIdx.x == 0) {
    int rank = -1;
    if ((int)xcc_id() == want) { unsigned t = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((int)t < W * NG) rank = (int)t; }
    s_rank = rank; s_fail = 0;
  }
  __syncthreads();
  if (s_rank < 0) return;
  const int grp = s_rank / W, wg = s_rank % W;
  unsigned* bar = sync + 16 + 16 * grp;                              // one 64-byte line per group
  double* gd = data + (size_t)grp * W * 512;
  if (threadIdx.x == 0) { long sp = 0; while (ldu(sync) < (unsigned)(W * NG)) { __builtin_amdgcn_s_sleep(2); if (++sp > (1L << 20)) { s_fail = 1; break; } } }
  __syncthreads();
  if (s_fail) { if (threadIdx.x == 0) atomicAdd(errs + 1, 1); return; }
  unsigned target = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    for (int half = 0; half < 2; ++half) {
      if (half == 0) gd[(size_t)wg * 512 + threadIdx.x] = (double)(r * 1000 + wg);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      target += W;
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long sp = 0;
        while (ldu(bar) < target) { if (++sp > (1L << 22)) { s_fail = 1; break; } }
      }
      __syncthreads();
      if (s_fail) break;
      if (half == 0) {
        const int nb = (wg + 1) % W;
        const double v = __hip_atomic_load(gd + (size_t)nb * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (double)(r * 1000 + nb)) atomicAdd(errs, 1);
      }
    }
    if (s_fail) break;
  }
  if (threadIdx.x == 0) { out[s_rank] = wall_clock64() - t0; if (s_fail) atomicAdd(errs + 1, 1); }
}

// ---- (B) whole chip: G workgroups (one per CU), vector of n doubles published in slices, gathered with a pseudo-random pattern
// mode 0: flat counter; mode 1: per-XCD counters (lines 16 * (1 + x)) + global counter (line 0)
__global__ __launch_bounds__(256) void k_chip_round(unsigned* sync, int G, int mode, int rounds, int n, int gathers, double* vec, unsigned long long* out, int* errs, int fence) {
  __shared__ int s_fail;
  __shared__ unsigned s_x;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) { s_fail = 0; s_x = xcc_id(); }
  __syncthreads();
  const unsigned x = s_x;
  // rendezvous: all G workgroups must be resident (G <= number of CUs)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sync + 15 * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long sp = 0;
    while (ldu(sync + 15 * 16) < (unsigned)G) { __builtin_amdgcn_s_sleep(2); if (++sp > (1L << 21)) { s_fail = 1; break; } }
  }
  __syncthreads();
  if (s_fail) { if (threadIdx.x == 0) atomicAdd(errs + 1, 1); return; }
  const int per = (n + G - 1) / G;
  const int lo = b * per, hi = min(n, lo + per);
  unsigned tgt_flat = 0, tgt_x = 0, tgt_g = 0;
  const int per_x = G / 8;
  unsigned long long t0 = wall_clock64();
  unsigned h = 2654435761u * (unsigned)(b * 256 + threadIdx.x) + 12345u;
  for (int r = 0; r < rounds; ++r) {
    for (int i = lo + threadIdx.x; i < hi; i += 256) vec[i] = (double)(r + 1) * 0.5 + (double)i;       // publish the slice
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      long sp = 0;
      if (mode == 0) {
        tgt_flat += G;
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ldu(sync) < tgt_flat) { if (++sp > (1L << 22)) { s_fail = 1; break; } }
      } else {
        tgt_x += per_x; tgt_g += 8;
        const unsigned old = __hip_atomic_fetch_add(sync + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == tgt_x) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // last of this XCD
        while (ldu(sync) < tgt_g) { if (++sp > (1L << 22)) { s_fail = 1; break; } }
      }
    }
    __syncthreads();
    if (s_fail) break;
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double acc = 0.0;
    for (int g = 0; g < gathers; ++g) {
      h = h * 1664525u + 1013904223u;
      const int i = (int)(h % (unsigned)n);
      const double v = fence ? vec[i] : __hip_atomic_load(vec + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (double)(r + 1) * 0.5 + (double)i) acc += 1.0;
    }
    if (acc != 0.0) atomicAdd(errs, 1);
    // (a second barrier would follow in a real iteration before the slice is overwritten; here the next round's values differ, and a reader that is
    //  still gathering when a fast writer overwrites would count as stale: so keep one more barrier, as the real kernel would)
    __syncthreads();
    if (threadIdx.x == 0) {
      long sp = 0;
      if (mode == 0) {
        tgt_flat += G;
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ldu(sync) < tgt_flat) { if (++sp > (1L << 22)) { s_fail = 1; break; } }
      } else {
        tgt_x += per_x; tgt_g += 8;
        const unsigned old = __hip_atomic_fetch_add(sync + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == tgt_x) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ldu(sync) < tgt_g) { if (++sp > (1L << 22)) { s_fail = 1; break; } }
      }
    }
    __syncthreads();
    if (s_fail) break;
  }
  if (threadIdx.x == 0) { out[b] = wall_clock64() - t0; if (s_fail) atomicAdd(errs + 1, 1); }
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("device %s, CUs %d\n", prop.name, prop.multiProcessorCount);
  unsigned* sync; double* data; unsigned long long* out; int* errs;
  hipMalloc(&sync, 4096 * 4); hipMalloc(&data, 64 * 8 * 512 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&errs, 8);
  const int rounds = 400;
  printf("(A) groups of W workgroups on one XCD, NG groups at once; a round = store, barrier, neighbour's read, barrier\n");
  for (int W : {2, 4, 8}) for (int NG : {1, 4}) {
    if (W * NG > 32) continue;
    hipMemset(sync, 0, 4096 * 4); hipMemset(errs, 0, 8); hipMemset(out, 0, 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_group_barrier, dim3(8 * 2 * W * NG), dim3(512), 0, 0, sync, W, NG, 0, rounds, data, out, errs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int he[2]; unsigned long long ho[64];
    hipMemcpy(he, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 64 * 8, hipMemcpyDeviceToHost);
    printf("  W=%d NG=%d: stale %d timeouts %d, %.3f us per barrier (in-kernel wall clock of rank 0: %.3f us)\n", W, NG, he[0], he[1], ms * 1e3 / (2 * rounds),
           (double)ho[0] / 100.0 / (2 * rounds));
  }
  printf("(B) whole chip, G workgroups x 256 threads; a round = publish slice of n = 50000 doubles, barrier, gather, barrier\n");
  double* vec; hipMalloc(&vec, 50000 * 8);
  for (int G : {64, 128, 256}) for (int mode : {0, 1}) for (int fence : {1, 0}) for (int gathers : {0, 11}) {
    hipMemset(sync, 0, 4096 * 4); hipMemset(errs, 0, 8); hipMemset(out, 0, 1024 * 8); hipMemset(vec, 0, 50000 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chip_round, dim3(G), dim3(256), 0, 0, sync, G, mode, rounds, 50000, gathers, vec, out, errs, fence);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int he[2]; unsigned long long ho[4];
    hipMemcpy(he, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 4 * 8, hipMemcpyDeviceToHost);
    printf("  G=%3d %s %s gathers=%2d: stale-read workgroup-rounds %d, timeouts %d, %.3f us per round = %.3f us per barrier incl. publish / gather\n", G,
           mode ? "per-XCD + global counter" : "flat counter           ", fence ? "release/acquire fences" : "no fences (sc1 loads) ", gathers, he[0], he[1],
           ms * 1e3 / rounds, ms * 1e3 / (2 * rounds));
  }
  return 0;
}
