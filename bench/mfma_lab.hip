// Lab: sustained v_mfma_f64_16x16x4_f64 rate of one MI355X, and how it degrades with the LDS fragment reads of the symmetric-product
// kernels.  Variants: pure MFMA (9 independent accumulators, register operands), MFMA + per-k-step LDS fragment reads into the SAME
// registers (the compiler's schedule in k_symm_gemm), MFMA + LDS reads double-buffered in registers (reads of step k+1 issued
// before the MFMAs of step k).  build: hipcc --offload-arch=gfx950 -O3 -o mfma_lab mfma_lab.hip ; run: ./mfma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k_lab(double* out, int steps) {
  __shared__ double lds[2 * 16 * 112 * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 16 * 112 * 2; i += 256) lds[i] = 1e-3 * (i & 15);
  __syncthreads();
  v4d acc[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) acc[a][b] = v4d{0, 0, 0, 0};
  const int fa = 48 * (wv & 1) + (lane & 15), fb = 48 * (wv >> 1) + (lane & 15), fk = lane >> 4;
  const double* As = lds;
  const double* Bs = lds + 16 * 112 * 2;
  double av[3] = {1.0 + lane, 2.0, 3.0}, bv[3] = {0.5, 0.25 + lane, 0.125};
  if (MODE == 0) {
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[a][b] = MFMA(av[a], bv[b], acc[a][b]);
    }
  } else if (MODE == 1) {
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double* ap = As + buf * 16 * 112 + (ks * 4 + fk) * 112 + fa;
        const double* bp = Bs + buf * 16 * 112 + (ks * 4 + fk) * 112 + fb;
#pragma unroll
        for (int a = 0; a < 3; ++a) { av[a] = ap[16 * a]; bv[a] = bp[16 * a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[a][b] = MFMA(av[a], bv[b], acc[a][b]);
      }
    }
  } else {
    double an[3], bn[3];
    {
      const double* ap = As + fk * 112 + fa;
      const double* bp = Bs + fk * 112 + fb;
#pragma unroll
      for (int a = 0; a < 3; ++a) { av[a] = ap[16 * a]; bv[a] = bp[16 * a]; }
    }
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kn = (ks + 1) & 3, bufn = (ks == 3) ? (buf ^ 1) : buf;
        const double* ap = As + bufn * 16 * 112 + (kn * 4 + fk) * 112 + fa;
        const double* bp = Bs + bufn * 16 * 112 + (kn * 4 + fk) * 112 + fb;
#pragma unroll
        for (int a = 0; a < 3; ++a) { an[a] = ap[16 * a]; bn[a] = bp[16 * a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[a][b] = MFMA(av[a], bv[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 3; ++a) { av[a] = an[a]; bv[a] = bn[a]; }
      }
    }
  }
  double sacc = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int MODE>
static void run(const char* name, int blocks, int threads, double* out) {
  const int steps = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_lab<MODE>, dim3(blocks), dim3(threads), 0, 0, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_lab<MODE>, dim3(blocks), dim3(threads), 0, 0, out, steps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * steps * 36.0 * 2048.0;
  printf("%-44s blocks %4d x %3d threads: %8.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz, %d waves/SIMD)\n", name, blocks, threads, ms,
         flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)blocks * (threads / 64) * steps * 36.0 / 1024.0), blocks * (threads / 64) / 1024);
}

int main() {
  double* out; hipMalloc(&out, 8 * 2048 * 256);
  for (int wps = 1; wps <= 2; ++wps) {
    run<0>("pure MFMA, register operands", 256 * wps, 256, out);
    run<1>("MFMA + LDS fragment reads, same registers", 256 * wps, 256, out);
    run<2>("MFMA + LDS reads one k-step ahead", 256 * wps, 256, out);
  }
  return 0;
}
