// xcd_lab.hip -- lab: where do the blocks of a launch land (HW_REG_XCC_ID census), and what does an arrival-counter barrier among the
// workgroups of ONE XCD cost (plain stores + sc1 loads, no fences)?   hipcc --offload-arch=gfx950 -O3 bench/xcd_lab.hip -o bench/xcd_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>

__global__ void k_census(unsigned* raw, unsigned long long* clk) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    raw[blockIdx.x] = x;
    clk[blockIdx.x] = wall_clock64();
  }
}

__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// W participants on XCD `want` do `rounds` barriers; each round every participant stores a value that its neighbour checks after the barrier
__global__ __launch_bounds__(1024) void k_barrier(unsigned* sync, int W, int want, int rounds, double* data, unsigned long long* out, int* errs) {
  __shared__ int s_rank, s_fail;
  if (threadIdx.x == 0) {
    int rank = -1;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if ((int)(x & 0xF) == want) { unsigned t = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((int)t < W) rank = (int)t; }
    s_rank = rank; s_fail = 0;
  }
  __syncthreads();
  const int wg = s_rank;
  if (wg < 0) return;
  if (threadIdx.x == 0) { long sp = 0; while (ldu(sync) < (unsigned)W) { __builtin_amdgcn_s_sleep(2); if (++sp > (1L << 20)) { s_fail = 1; break; } } }
  __syncthreads();
  if (s_fail) { if (threadIdx.x == 0) atomicAdd(errs + 1, 1); return; }
  unsigned bar = 0;
  unsigned long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    data[(size_t)wg * 1024 + threadIdx.x] = (double)(r * 1000 + wg);            // plain store
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    bar += W;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long sp = 0;
      while (ldu(sync + 1) < bar) { __builtin_amdgcn_s_sleep(1); if (++sp > (1L << 22)) { s_fail = 1; break; } }
    }
    __syncthreads();
    if (s_fail) break;
    const int nb = (wg + 1) % W;
    const double v = __hip_atomic_load(data + (size_t)nb * 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load
    if (v != (double)(r * 1000 + nb)) atomicAdd(errs, 1);
    // second barrier so that nobody overwrites before everybody has read
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    bar += W;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long sp = 0;
      while (ldu(sync + 1) < bar) { __builtin_amdgcn_s_sleep(1); if (++sp > (1L << 22)) { s_fail = 1; break; } }
    }
    __syncthreads();
    if (s_fail) break;
  }
  if (threadIdx.x == 0) { out[wg] = wall_clock64() - t0; if (s_fail) atomicAdd(errs + 1, 1); }
}

int main() {
  const int G = 1024;
  unsigned* raw; unsigned long long* clk;
  hipMalloc(&raw, G * 4); hipMalloc(&clk, G * 8);
  for (int threads : {256, 1024}) {
    hipLaunchKernelGGL(k_census, dim3(G), dim3(threads), 0, 0, raw, clk);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), raw, G * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> hist; int rr_ok = 0;
    for (int b = 0; b < G; ++b) { hist[h[b]]++; if ((int)(h[b] & 0xF) == b % 8) rr_ok++; }
    printf("census threads=%d: raw values:", threads);
    for (auto& kv : hist) printf(" 0x%x:%d", kv.first, kv.second);
    printf("  | block b on XCD b%%8: %d / %d | first 16:", rr_ok, G);
    for (int b = 0; b < 16; ++b) printf(" %x", h[b]);
    printf("\n");
  }
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("CUs %d, wall clock rate %d kHz\n", prop.multiProcessorCount, prop.clockRate);
  unsigned* sync; double* data; unsigned long long* out; int* errs;
  hipMalloc(&sync, 16); hipMalloc(&data, 64 * 1024 * 8 * 2); hipMalloc(&out, 128 * 8); hipMalloc(&errs, 8);
  for (int W : {16, 32, 48, 64}) {
    for (int mult : {2, 4}) {
      hipMemset(sync, 0, 16); hipMemset(errs, 0, 8); hipMemset(out, 0, 128 * 8);
      const int rounds = 200;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_barrier, dim3(8 * mult * W), dim3(1024), 0, 0, sync, W, 0, rounds, data, out, errs);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int he[2]; unsigned hs[4]; unsigned long long ho[128];
      hipMemcpy(he, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sync, 16, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 128 * 8, hipMemcpyDeviceToHost);
      printf("W=%2d grid=%4d: tickets on XCD0 %u, stale reads %d, timeouts %d, kernel %.1f us => %.2f us per barrier (wall clocks/barrier %.0f at 100 MHz)\n", W, 8 * mult * W,
             hs[0], he[0], he[1], ms * 1e3, ms * 1e3 / (2 * rounds), (double)ho[0] / (2 * rounds));
    }
  }
  return 0;
}
