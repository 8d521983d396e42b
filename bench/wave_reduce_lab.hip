// wave_reduce_lab.hip -- the DPP form of wave_sum (csrc/device_utils.h: xor 32 / 16 through v_permlane32_swap / v_permlane16_swap, then row_ror:8, row_ror:4, quad_perm) against the
// six-step __shfl_xor butterfly it replaces: same bits in every lane, for 4096 waves of random data over 12 decades.
//   hipcc --offload-arch=gfx950 -O3 bench/wave_reduce_lab.hip -o bench/wave_reduce_lab && bench/wave_reduce_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
template <int CTRL> __device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <bool ROWS16> __device__ __forceinline__ void permlane_pair(double v, double& a, double& b) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto r = ROWS16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto q = ROWS16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double((int)q[0], (int)r[0]); b = __hiloint2double((int)q[1], (int)r[1]);
}
__device__ double ws_new(double v) {
  double a, b;
  permlane_pair<false>(v, a, b); v = a + b;
  permlane_pair<true>(v, a, b); v = a + b;
  v += dpp_move<0x128>(v); v += dpp_move<0x124>(v); v += dpp_move<0x4E>(v); v += dpp_move<0xB1>(v); return v; }
__device__ double ws_old(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__global__ void k(const double* in, double* a, double* b) { int i = blockIdx.x * 64 + threadIdx.x; a[i] = ws_new(in[i]); b[i] = ws_old(in[i]); }
__global__ void ktime(double* out, long long* cyc) {
  double v = threadIdx.x * 1e-3 + 1.0, w = v;
  long long t0 = clock64();
  for (int i = 0; i < 1000; ++i) v = ws_new(v) * 1e-2 + 1.0;
  long long t1 = clock64();
  for (int i = 0; i < 1000; ++i) w = ws_old(w) * 1e-2 + 1.0;
  long long t2 = clock64();
  out[threadIdx.x] = v - w;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  const int N = 64 * 4096; double *h = (double*)malloc(N * 8), *d, *a, *b; srand(1);
  for (int i = 0; i < N; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10.0, rand() % 12 - 6);
  hipMalloc(&d, N * 8); hipMalloc(&a, N * 8); hipMalloc(&b, N * 8); hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
  k<<<N / 64, 64>>>(d, a, b); double *ha = (double*)malloc(N * 8), *hb = (double*)malloc(N * 8);
  hipMemcpy(ha, a, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, N * 8, hipMemcpyDeviceToHost);
  int bad = 0, lanebad = 0; for (int i = 0; i < N; ++i) { if (memcmp(&ha[i], &hb[i], 8)) ++bad; if (memcmp(&ha[i], &ha[i & ~63], 8)) ++lanebad; }
  long long* cyc; hipMalloc(&cyc, 16); ktime<<<1, 64>>>(a, cyc); long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
  printf("one dependent wave_sum (+ 1 fma), single wave, shader cycles (clock64): permlane/DPP form %.2f, __shfl_xor form %.2f per reduction\n", hc[0] / 1000.0, hc[1] / 1000.0);
  printf("mismatches vs shfl butterfly: %d, lanes differing within a wave: %d\n", bad, lanebad); return bad || lanebad;
}
