// Lab: can the host keep a stream of 4-8 us dependent kernels fed, and what does a hipGraph replay of the same chain cost?
// Kernel: one workgroup-grid of `blocks` x 256 threads that spins for `cycles` shader clocks (s_memtime), then writes one word.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_lab launch_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void k_spin(long long cycles, int* out, const int* gate) {
  if (gate && *gate) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* out; hipMalloc(&out, 64); hipMemset(out, 0, 64);
  hipStream_t st; hipStreamCreate(&st);
  const int N = 20000;
  for (int blocks : {196, 922}) {
    for (long long us : {0LL, 2LL, 4LL, 8LL}) {
      const long long cycles = us * 100;   // wall_clock64 = 100 MHz constant clock (s_memrealtime): 100 ticks per us
      hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, cycles, out, (const int*)nullptr); hipStreamSynchronize(st);
      double t0 = now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, cycles, out, (const int*)nullptr);
      double t1 = now();
      hipStreamSynchronize(st);
      double t2 = now();
      printf("stream  blocks %4d spin %lld us: host enqueue %.2f us/launch, end-to-end %.2f us/launch\n", blocks, us, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
      // the same chain as a graph of 64 kernel nodes, replayed
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, cycles, out, (const int*)nullptr);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      t0 = now();
      for (int i = 0; i < N / 64; ++i) hipGraphLaunch(ge, st);
      t1 = now();
      hipStreamSynchronize(st);
      t2 = now();
      printf("graph64 blocks %4d spin %lld us: host enqueue %.2f us/kernel, end-to-end %.2f us/kernel\n", blocks, us, 1e6 * (t1 - t0) / (N / 64 * 64), 1e6 * (t2 - t0) / (N / 64 * 64));
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
