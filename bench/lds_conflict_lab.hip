// lds_conflict_lab.hip -- what does ONE LDS read of a wave cost inside a 512-thread workgroup (8 waves, the shape of the register batch kernel),
// depending on the address pattern?  The sparse passes of k_batch_admm_reg issue three LDS reads per nonzero (value b64, index u16 / packed pair b32,
// gathered operand b64); their cost per wave-instruction decides whether a wave-trip-major (conflict-free) layout of the values / indices would pay.
// Patterns (address of lane l at step t):
//   b64 linear      8 * (64 t + l)                 consecutive lanes -> consecutive 8-byte slots (what a wave-trip-major layout gives)
//   b64 stride10    8 * (10 l + t)                 row-major rows of 10 nonzeros, one row per lane (values of an unsorted CSR pass)
//   b64 random      8 * hash(l, t) mod NSLOT       (gathers; values of length-sorted rows)
//   b64 colored     random, but the 32 lanes of a half-wave fall on 32 different bank pairs (slot mod 32 = a permutation of the lane index)
//   b64 colored16   the same per group of 16 lanes
//   b32 / u16 linear, stride10, random             (index reads)
// Every thread holds 16 precomputed addresses per pattern and issues NREP reads over them (independent, accumulated); cycles = clock64 of thread 0 around the loop, all 8 waves
// running the same loop.  Output: cycles per wave-instruction (wall of the workgroup / (NREP * 8 waves)).
//   hipcc --offload-arch=gfx950 -O3 bench/lds_conflict_lab.hip -o bench/lds_conflict_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define NSLOT 8192          // 64 KB of doubles
#define NREP 2048
__device__ __forceinline__ unsigned hsh(unsigned a, unsigned b) { unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h; }

template <int PAT>
__device__ __forceinline__ int slot_of(int l, int t) {
  if (PAT == 0) return (64 * t + l) % NSLOT;
  if (PAT == 1) return (10 * l + 10 * 64 * (t / 10) + t % 10) % NSLOT;
  if (PAT == 2) return (int)(hsh((unsigned)l, (unsigned)t) % NSLOT);
  if (PAT == 3) { const int half = l & 31; const int bank = (half * 7 + t * 5) & 31; return (int)(((hsh((unsigned)l, (unsigned)t) % (NSLOT / 32)) * 32 + bank)); }
  { const int q = l & 15; const int bank = ((q * 7 + t * 5) & 15) + 16 * ((l >> 4) & 1); return (int)(((hsh((unsigned)l, (unsigned)t) % (NSLOT / 32)) * 32 + bank)); }
}

template <int PAT, int WIDTH>    // WIDTH 8: double, 4: unsigned, 2: unsigned short; WIDTH 0: the loop without the LDS read (overhead)
__global__ __launch_bounds__(512) void k_lab(double* out, unsigned long long* cyc) {
  __shared__ double lds[NSLOT];
  for (int i = threadIdx.x; i < NSLOT; i += 512) lds[i] = (double)(i & 1023);
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  // 16 addresses per thread, computed once (the pattern's steps t = 0 .. 15, shifted per wave), reused NREP / 16 times: no address arithmetic in the loop
  int a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int t = k + 16 * w;
    int s = slot_of<PAT>(l, t);
    if (WIDTH == 4 && PAT <= 1) s = (PAT == 0 ? 64 * t + l : 10 * l + 10 * 64 * (t / 10) + t % 10) % (2 * NSLOT);
    if (WIDTH == 4 && PAT == 2) s = (int)(hsh((unsigned)l, (unsigned)t) % (2 * NSLOT));
    if (WIDTH == 2 && PAT <= 1) s = (PAT == 0 ? 64 * t + l : 10 * l + 10 * 64 * (t / 10) + t % 10) % (4 * NSLOT);
    if (WIDTH == 2 && PAT == 2) s = (int)(hsh((unsigned)l, (unsigned)t) % (4 * NSLOT));
    a[k] = s;
  }
  double acc = 0.0; unsigned acc_u = 0;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int r = 0; r < NREP / 16; ++r) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (WIDTH == 8) acc += lds[a[k]];
      else if (WIDTH == 4) acc_u += reinterpret_cast<const unsigned*>(lds)[a[k]];
      else if (WIDTH == 2) acc_u += reinterpret_cast<const unsigned short*>(lds)[a[k]];
      else acc_u += (unsigned)a[k];
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc + (double)acc_u;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static double run(K kern, double* out, unsigned long long* cyc) {
  hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, out, cyc);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, out, cyc);
  (void)hipDeviceSynchronize();
  unsigned long long c = 0;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  return (double)c;
}

int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 8 * 8); (void)hipMalloc(&cyc, 64);
  const double nwi = (double)NREP * 8.0;     // wave-instructions of the workgroup
  const char* names[] = {"linear", "stride10", "random", "colored32", "colored16"};
  double base[5] = {run(k_lab<0, 0>, out, cyc), run(k_lab<1, 0>, out, cyc), run(k_lab<2, 0>, out, cyc), run(k_lab<3, 0>, out, cyc), run(k_lab<4, 0>, out, cyc)};
  double b64[5] = {run(k_lab<0, 8>, out, cyc), run(k_lab<1, 8>, out, cyc), run(k_lab<2, 8>, out, cyc), run(k_lab<3, 8>, out, cyc), run(k_lab<4, 8>, out, cyc)};
  double b32[3] = {run(k_lab<0, 4>, out, cyc), run(k_lab<1, 4>, out, cyc), run(k_lab<2, 4>, out, cyc)};
  double u16[3] = {run(k_lab<0, 2>, out, cyc), run(k_lab<1, 2>, out, cyc), run(k_lab<2, 2>, out, cyc)};
  printf("one workgroup of 512 threads (8 waves), %d reads per thread; shader-clock cycles per wave-instruction (in brackets: after subtracting the address arithmetic alone)\n", NREP);
  for (int p = 0; p < 5; ++p) printf("  ds_read_b64  %-10s %6.2f  (%6.2f)\n", names[p], b64[p] / nwi, (b64[p] - base[p]) / nwi);
  for (int p = 0; p < 3; ++p) printf("  ds_read_b32  %-10s %6.2f  (%6.2f)\n", names[p], b32[p] / nwi, (b32[p] - base[p]) / nwi);
  for (int p = 0; p < 3; ++p) printf("  ds_read_u16  %-10s %6.2f  (%6.2f)\n", names[p], u16[p] / nwi, (u16[p] - base[p]) / nwi);
  return 0;
}
