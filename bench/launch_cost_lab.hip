// Lab: host cost and device pace of N small dependent kernel launches, direct vs one hipGraph launch of the captured chain.
// build: hipcc --offload-arch=gfx950 -O2 bench/launch_cost_lab.hip -o bench/_lab/launch_cost_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { const double* a; const int* b; const int* c; const int* d; const int* e; int n, m, k, l; long long x, y; };   // ~ a CsrView passed by value
__global__ void k_small(double* p, int iters, Big v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s = p[i];
  for (int t = 0; t < iters; ++t) s = s * 1.0000001 + 1e-9;
  p[i] = s + (double)v.n * 0.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double* d; hipMalloc(&d, 8 * 256 * 200);
  hipMemset(d, 0, 8 * 256 * 200);
  hipStream_t st; hipStreamCreate(&st);
  Big v{}; v.n = 3;
  for (int iters : {0, 2000}) {                      // ~1.5 us and ~10 us kernels
    for (int chain : {16, 64, 256}) {
      const int reps = 4096 / chain * 4;
      // direct
      for (int w = 0; w < 2; ++w) {
        hipStreamSynchronize(st);
        const double t0 = now();
        for (int r = 0; r < reps; ++r) for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_small, dim3(200), dim3(256), 0, st, d, iters, v);
        const double t1 = now();
        hipStreamSynchronize(st);
        const double t2 = now();
        if (w) printf("iters %4d chain %3d direct: host %.2f us/launch, total %.2f us/launch\n", iters, chain, 1e6 * (t1 - t0) / (reps * chain), 1e6 * (t2 - t0) / (reps * chain));
      }
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_small, dim3(200), dim3(256), 0, st, d, iters, v);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int w = 0; w < 2; ++w) {
        hipStreamSynchronize(st);
        const double t0 = now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
        const double t1 = now();
        hipStreamSynchronize(st);
        const double t2 = now();
        if (w) printf("iters %4d chain %3d graph : host %.2f us/launch, total %.2f us/launch\n", iters, chain, 1e6 * (t1 - t0) / (reps * chain), 1e6 * (t2 - t0) / (reps * chain));
      }
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
