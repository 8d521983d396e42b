// xcd_resident_lab.hip -- lab for VERDICT r04 item 8 (the question DESIGN section 10 left open): can a group of cones' work matrices STAY in one XCD's
// 4 MB L2 across the products of a sign-iteration step, if the step is one persistent per-XCD kernel with a software barrier between products?
//
// What decides it: what survives in the L2 across an in-kernel barrier, and at which scope the barrier's fences must be.  On this part the eight L2s
// are not coherent with each other; an AGENT-scope acquire is `buffer_inv sc1`, a release `buffer_wbl2 sc1`.  Inside ONE XCD the L2 is the
// coherence point of its 32 CUs: a CU's L1 is write-through, so another CU of the same XCD sees a store as soon as it has reached the L2 -- provided
// it does not read a stale line of its own L1: `buffer_inv sc0` (workgroup scope: L1 only) is all the "acquire" the exchange needs.
//
// The lab: 32 workgroups per XCD (one per CU, XCD identified by HW_REG_XCC_ID), each XCD on its OWN region of S bytes.  Per round every workgroup
// streams the WHOLE region (the operand panels every tile of a product reads), checks that it holds the value of the previous round, then overwrites
// its 1/32 slice with the next value (the product's output) and passes a barrier among the 32 workgroups of its XCD.  Variants of the barrier's fences:
//   inv_sc0  : s_waitcnt vmcnt(0) | counter barrier | buffer_inv sc0                       (in-XCD exchange, L2 contents untouched)
//   agent    : release fence (agent) | counter barrier | acquire fence (agent)              (what cross-XCD visibility would need)
//   none     : s_waitcnt vmcnt(0) | counter barrier                                        (expected: stale reads out of the L1)
//   launches : no in-kernel barrier, one kernel launch per round                            (today's structure: the L2s are invalidated at every boundary)
// Reported per variant and S: microseconds per round, stale reads, read bandwidth per XCD (32 x S / t) and over the chip.
//   hipcc --offload-arch=gfx950 -O3 bench/xcd_resident_lab.hip -o bench/xcd_resident_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define WG_PER_XCD 32
#define BS 256
__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xF; }

// sync layout (unsigned, 16 per 64-byte line): line x = arrival counter of XCD x (x < 8), line 8 + x = rank ticket of XCD x
// variant: 0 inv_sc0, 1 agent, 2 none, 3 single round (launch-per-round mode: round index passed in, no barrier)
__global__ __launch_bounds__(BS) void k_resident(unsigned* sync, double* buf, long long elems_per_xcd, int rounds, int variant, int round0,
                                                 unsigned long long* cycles, unsigned long long* stale) {
  __shared__ int s_rank, s_x, s_fail;
  __shared__ double red[BS / 64];
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    const unsigned t = __hip_atomic_fetch_add(sync + 16 * (8 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_x = (int)x; s_rank = (variant == 3) ? (int)(t % WG_PER_XCD) : ((t < WG_PER_XCD) ? (int)t : -1); s_fail = 0;    // (launch-per-round: the tickets run on, 32 arrivals per XCD and launch)
  }
  __syncthreads();
  if (s_rank < 0) return;
  const int x = s_x, rank = s_rank;
  double* reg = buf + (size_t)x * (size_t)elems_per_xcd;
  const long long per = elems_per_xcd / WG_PER_XCD;
  unsigned* bar = sync + 16 * x;
  unsigned target = 0;
  unsigned long long bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    const double expect = (double)(round0 + r);
    // read the whole region: 16-byte loads, every workgroup of the XCD reads everything
    const double2* p2 = reinterpret_cast<const double2*>(reg);
    double acc = 0.0;
    for (long long i = threadIdx.x; i < elems_per_xcd / 2; i += BS) { const double2 v = p2[i]; acc += (v.x - expect) * (v.x - expect) + (v.y - expect) * (v.y - expect); }
    // block sum of the squared deviations: 0 iff every element held `expect`
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < BS / 64; ++w) t += red[w]; if (t != 0.0) bad += 1; }
    if (variant == 3) {                                      // launch-per-round: write the slice and leave
      for (long long i = threadIdx.x; i < per; i += BS) reg[(size_t)rank * per + i] = expect + 1.0;
      break;
    }
    // every reader of the XCD must be done before anybody overwrites: barrier 1 (no data exchange: no fence needed)
    __syncthreads();
    target += WG_PER_XCD;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long sp = 0; while (ldu(bar) < target) { if (++sp > (1L << 24)) { s_fail = 1; break; } }
    }
    __syncthreads();
    if (s_fail) break;
    for (long long i = threadIdx.x; i < per; i += BS) reg[(size_t)rank * per + i] = expect + 1.0;
    if (variant == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += WG_PER_XCD;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long sp = 0; while (ldu(bar) < target) { if (++sp > (1L << 24)) { s_fail = 1; break; } }
    }
    __syncthreads();
    if (s_fail) break;
    if (variant == 0) asm volatile("buffer_inv sc0" ::: "memory");
    else if (variant == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (threadIdx.x == 0) { cycles[x * WG_PER_XCD + rank] = wall_clock64() - t0; stale[x * WG_PER_XCD + rank] = bad + (s_fail ? (1ull << 40) : 0); }
}

__global__ void k_fill(double* buf, long long n, double v) { for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) buf[i] = v; }

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("device %s, %d CUs, wall clock %d kHz, L2 %d KB (per XCD), rounds %d\n", prop.name, prop.multiProcessorCount, wall_khz, prop.l2CacheSize / 1024, rounds);
  unsigned* sync; unsigned long long *cyc, *stale;
  hipMalloc(&sync, 64 * 16 * sizeof(unsigned));
  hipMalloc(&cyc, 256 * sizeof(unsigned long long)); hipMalloc(&stale, 256 * sizeof(unsigned long long));
  const double mbs[] = {0.5, 1, 2, 3, 4, 6, 8, 16};
  const char* names[] = {"inv_sc0", "agent", "none", "launches"};
  for (double mb : mbs) {
    long long elems = (long long)(mb * 1024 * 1024 / 8); elems = elems / (WG_PER_XCD * 2) * (WG_PER_XCD * 2);
    double* buf; hipMalloc(&buf, sizeof(double) * elems * 8);
    for (int variant = 0; variant < 4; ++variant) {
      hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, buf, elems * 8, 0.0);
      hipMemset(cyc, 0, 256 * 8); hipMemset(stale, 0, 256 * 8);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0.f;
      if (variant < 3) {
        hipMemset(sync, 0, 64 * 16 * sizeof(unsigned));
        hipEventRecord(e0, 0);
        // 2 x 256 workgroups so that every XCD certainly receives >= 32 (the surplus leaves at once); the CU holds one 256-thread workgroup of this size
        hipLaunchKernelGGL(k_resident, dim3(256), dim3(BS), 0, 0, sync, buf, elems, rounds, variant, 0, cyc, stale);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      } else {
        hipMemset(sync, 0, 64 * 16 * sizeof(unsigned));
        hipEventRecord(e0, 0);
        for (int r = 0; r < rounds; ++r) {
          hipLaunchKernelGGL(k_resident, dim3(256), dim3(BS), 0, 0, sync, buf, elems, 1, 3, r, cyc, stale);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> hc(256), hs(256);
      hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), stale, 256 * 8, hipMemcpyDeviceToHost);
      unsigned long long st = 0, hung = 0; int active = 0;
      for (int i = 0; i < 256; ++i) { if (hc[i]) active += 1; st += hs[i] & ((1ull << 40) - 1); hung += hs[i] >> 40; }
      const double us = 1e3 * ms / rounds;
      const double bw_xcd = (double)WG_PER_XCD * elems * 8.0 / (us * 1e-6) / 1e12;
      printf("S = %5.1f MB per XCD  %-9s %8.2f us per round  read %6.2f TB/s per XCD (%6.1f TB/s chip)  stale workgroup-rounds %llu  hung %llu  active workgroups %d\n",
             mb, names[variant], us, bw_xcd, 8.0 * bw_xcd, st, hung, variant < 3 ? active : 256);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipFree(buf);
  }
  return 0;
}
