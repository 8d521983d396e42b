// spmv_lab.hip -- standalone SpMV kernel laboratory for gfx950 (not part of the library).
// Builds random CSR matrices with the statistics of BASELINE config 2 and times kernel variants back to back with HIP
// events; verifies every variant against a serial CPU loop.   hipcc --offload-arch=gfx950 -O3 bench/spmv_lab.hip -o spmv_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <random>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Csr { int nr, nc; std::vector<int> rp, col; std::vector<double> val; };

static Csr random_csr(int nr, int nc, long long nnz, unsigned seed) {
  std::mt19937_64 g(seed);
  std::vector<std::pair<long long, double>> e(nnz);
  std::normal_distribution<double> nd;
  for (long long k = 0; k < nnz; ++k) { long long i = g() % nr, j = g() % nc; e[k] = {i * (long long)nc + j, nd(g)}; }
  std::sort(e.begin(), e.end(), [](auto& a, auto& b) { return a.first < b.first; });
  Csr M; M.nr = nr; M.nc = nc; M.rp.assign(nr + 1, 0);
  for (long long k = 0; k < nnz; ++k) {
    if (k > 0 && e[k].first == e[k - 1].first) { M.val.back() += e[k].second; continue; }
    M.col.push_back((int)(e[k].first % nc)); M.val.push_back(e[k].second); M.rp[e[k].first / nc + 1]++;
  }
  for (int i = 0; i < nr; ++i) M.rp[i + 1] += M.rp[i];
  return M;
}

// ---------------------------------------------------------------------------------------------------------------------
// V0/V1: CSR-stream, tile T nonzeros per workgroup, one thread per row in phase 2 (sequential order)
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int T, bool NT>
__global__ __launch_bounds__(BS) void k_stream(const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                                               const int* __restrict__ rb, int nb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double lds[T];
  for (int k = blockIdx.x; k < nb; k += gridDim.x) {
    const int r0 = rb[k], r1 = rb[k + 1];
    const int nz0 = rp[r0], cnt = rp[r1] - nz0;
#pragma unroll
    for (int it = 0; it < T / BS; ++it) {
      const int i = it * BS + threadIdx.x;
      if (i < cnt) {
        int c; double a;
        if (NT) { c = __builtin_nontemporal_load(&col[nz0 + i]); a = __builtin_nontemporal_load(&val[nz0 + i]); }
        else { c = col[nz0 + i]; a = val[nz0 + i]; }
        lds[i] = a * x[c];
      }
    }
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += BS) {
      const int a = rp[r] - nz0, b = rp[r + 1] - nz0;
      double s = 0.0;
      for (int j = a; j < b; ++j) s += lds[j];
      y[r] = s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// V2: CSR-vector with L lanes per row (L = 2,4,8,16,32,64), no LDS
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int L, bool NT>
__global__ __launch_bounds__(BS) void k_vector(int nr, const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                                               const double* __restrict__ x, double* __restrict__ y) {
  const int gid = blockIdx.x * BS + threadIdx.x;
  const int lane = gid & (L - 1);
  const int nrg = (gridDim.x * BS) / L;
  for (int r = gid / L; r < nr; r += nrg) {
    const int a = rp[r], b = rp[r + 1];
    double s = 0.0;
    for (int j = a + lane; j < b; j += L) {
      int c; double v;
      if (NT) { c = __builtin_nontemporal_load(&col[j]); v = __builtin_nontemporal_load(&val[j]); }
      else { c = col[j]; v = val[j]; }
      s += v * x[c];
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) y[r] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// V3: CSR-stream, register-staged products, then LDS; phase 2 with G lanes per row (parallel short-row reduce)
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int T, int G>
__global__ __launch_bounds__(BS) void k_stream_g(const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                                                 const int* __restrict__ rb, int nb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double lds[T];
  for (int k = blockIdx.x; k < nb; k += gridDim.x) {
    const int r0 = rb[k], r1 = rb[k + 1];
    const int nz0 = rp[r0], cnt = rp[r1] - nz0;
#pragma unroll
    for (int it = 0; it < T / BS; ++it) {
      const int i = it * BS + threadIdx.x;
      if (i < cnt) lds[i] = val[nz0 + i] * x[col[nz0 + i]];
    }
    __syncthreads();
    const int lane = threadIdx.x & (G - 1);
    for (int r = r0 + threadIdx.x / G; r < r1; r += BS / G) {
      const int a = rp[r] - nz0, b = rp[r + 1] - nz0;
      double s = 0.0;
      for (int j = a + lane; j < b; j += G) s += lds[j];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) y[r] = s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// V4: panel-sorted CSR-stream.  Inside every tile the (col, val) pairs are stored sorted by column together with the slot
// `pos` they occupy in row order; lanes of a wave then gather neighbouring lines of x (duplicates coalesce), the products
// are scattered to lds[pos] and phase 2 is unchanged (row order => bit-exact sums).
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int T>
__global__ __launch_bounds__(BS) void k_stream_sorted(const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                                                      const unsigned short* __restrict__ pos, const int* __restrict__ rb, int nb,
                                                      const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double lds[T];
  for (int k = blockIdx.x; k < nb; k += gridDim.x) {
    const int r0 = rb[k], r1 = rb[k + 1];
    const int nz0 = rp[r0], cnt = rp[r1] - nz0;
#pragma unroll
    for (int it = 0; it < T / BS; ++it) {
      const int i = it * BS + threadIdx.x;
      if (i < cnt) lds[pos[nz0 + i]] = val[nz0 + i] * x[col[nz0 + i]];
    }
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += BS) {
      const int a = rp[r] - nz0, b = rp[r + 1] - nz0;
      double s = 0.0;
      for (int j = a; j < b; ++j) s += lds[j];
      y[r] = s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// V5: CSR-stream, NT tiles per workgroup, software pipelined: the (col,val) loads and x gathers of tile i+1 are issued
// before tile i is reduced from LDS (register double buffering), so the memory pipe never idles during phase 2.
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int T, int NT>
__global__ __launch_bounds__(BS) void k_stream_pipe(const int* __restrict__ rp, const int* __restrict__ col, const double* __restrict__ val,
                                                    const int* __restrict__ rb, int nb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double lds[T];
  constexpr int E = T / BS;
  const int k0 = blockIdx.x * NT;
  double pr[E];
  int cnt_next = 0, nz0_next = 0;
  auto fetch = [&](int k) {   // products of tile k into registers
    const int r0 = rb[k], r1 = rb[k + 1];
    nz0_next = rp[r0]; cnt_next = rp[r1] - nz0_next;
#pragma unroll
    for (int it = 0; it < E; ++it) {
      const int i = it * BS + threadIdx.x;
      pr[it] = (i < cnt_next) ? val[nz0_next + i] * x[col[nz0_next + i]] : 0.0;
    }
  };
  if (k0 < nb) fetch(k0);
  for (int t = 0; t < NT; ++t) {
    const int k = k0 + t;
    if (k >= nb) break;
    const int r0 = rb[k], r1 = rb[k + 1];
    const int nz0 = nz0_next, cnt = cnt_next;
#pragma unroll
    for (int it = 0; it < E; ++it) { const int i = it * BS + threadIdx.x; if (i < cnt) lds[i] = pr[it]; }
    __syncthreads();
    if (t + 1 < NT && k + 1 < nb) fetch(k + 1);          // loads of the next tile fly while this tile is reduced
    for (int r = r0 + threadIdx.x; r < r1; r += BS) {
      const int a = rp[r] - nz0, b = rp[r + 1] - nz0;
      double s = 0.0;
      for (int j = a; j < b; ++j) s += lds[j];
      y[r] = s;
    }
    __syncthreads();
  }
}

// ---- diagnostics: isolate streaming, gathering and launch costs ----------------------------------------------------
__global__ void k_empty() {}
template <int MODE>   // 0: stream val,col only   1: + coalesced x   2: + random gather restricted to 4096 entries   3: full gather
__global__ __launch_bounds__(256) void k_diag(long long nnz, int nc, const int* __restrict__ col, const double* __restrict__ val,
                                              const double* __restrict__ x, double* __restrict__ y) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (long long)gridDim.x * 256) {
    const int c = col[i]; const double a = val[i];
    double xv;
    if (MODE == 0) xv = (double)c;
    else if (MODE == 1) xv = x[i % nc];
    else if (MODE == 2) xv = x[c & 4095];
    else xv = x[c];
    acc += a * xv;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) y[(blockIdx.x * 256 + threadIdx.x) >> 6] = acc;
}
// gather only: no matrix stream at all (column index from a hash)
__global__ __launch_bounds__(256) void k_gather_only(long long nnz, int nc, const double* __restrict__ x, double* __restrict__ y) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (long long)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    acc += x[h % (unsigned)nc];
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) y[(blockIdx.x * 256 + threadIdx.x) >> 6] = acc;
}

// gather-only with different cache policies on the gathered load: 0 plain, 1 nontemporal, 2 sc1 (agent-scope relaxed atomic load)
template <int POL>
__global__ __launch_bounds__(256) void k_gather_pol(long long nnz, int nc, const int* __restrict__ col, const double* __restrict__ x, double* __restrict__ y) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (long long)gridDim.x * 256) {
    const int c = col[i];
    double v;
    if (POL == 0) v = x[c];
    else if (POL == 1) v = __builtin_nontemporal_load(&x[c]);
    else v = __hip_atomic_load(&x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) y[(blockIdx.x * 256 + threadIdx.x) >> 6] = acc;
}

static void build_rb(const Csr& M, int T, int rows_max, std::vector<int>& rb) {
  rb.clear(); rb.push_back(0);
  int r = 0;
  while (r < M.nr) {
    int r1 = r; long long cnt = 0;
    while (r1 < M.nr) {
      long long rn = M.rp[r1 + 1] - M.rp[r1];
      if (r1 > r && (cnt + rn > T || r1 - r >= rows_max)) break;
      cnt += rn; ++r1;
      if (cnt > T) break;
    }
    rb.push_back(r1); r = r1;
  }
}

struct Dev { int *rp, *col, *rb; double *val, *x, *y; int nb; };

template <class F>
static double time_it(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3 / reps;
}

static void check(const char* name, const Csr& M, const std::vector<double>& x, const double* dy, double us, double bytes) {
  std::vector<double> y(M.nr), ref(M.nr);
  CK(hipMemcpy(y.data(), dy, sizeof(double) * M.nr, hipMemcpyDeviceToHost));
  double maxerr = 0, exact = 1;
  for (int i = 0; i < M.nr; ++i) {
    double s = 0; for (int j = M.rp[i]; j < M.rp[i + 1]; ++j) s += M.val[j] * x[M.col[j]];
    maxerr = fmax(maxerr, fabs(s - y[i])); if (s != y[i]) exact = 0;
  }
  printf("  %-34s %8.2f us  %7.1f GB/s  maxerr %.1e %s\n", name, us, bytes / us / 1e3, maxerr, exact ? "(bit-exact)" : "");
}

static void run_matrix(const char* label, int nr, int nc, long long nnz, unsigned seed) {
  Csr M = random_csr(nr, nc, nnz, seed);
  const long long z = (long long)M.val.size();
  const double bytes = 12.0 * z + 4.0 * (nr + 1) + 8.0 * nc + 8.0 * nr;
  printf("%s: %d x %d, nnz %lld (%.1f/row), algorithmic bytes %.2f MB, roofline@8TB/s %.2f us\n", label, nr, nc, z, (double)z / nr,
         bytes / 1e6, bytes / 8e6);
  std::vector<double> x(nc); std::mt19937_64 g(7); std::normal_distribution<double> nd; for (auto& v : x) v = nd(g);
  Dev d;
  CK(hipMalloc(&d.rp, sizeof(int) * (nr + 1))); CK(hipMalloc(&d.col, sizeof(int) * z)); CK(hipMalloc(&d.val, sizeof(double) * z));
  CK(hipMalloc(&d.x, sizeof(double) * nc)); CK(hipMalloc(&d.y, sizeof(double) * nr)); CK(hipMalloc(&d.rb, sizeof(int) * (nr + 2)));
  CK(hipMemcpy(d.rp, M.rp.data(), sizeof(int) * (nr + 1), hipMemcpyHostToDevice));
  CK(hipMemcpy(d.col, M.col.data(), sizeof(int) * z, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.val, M.val.data(), sizeof(double) * z, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.x, x.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
  const int R = 200;
  std::vector<int> rb;
  {
    double us = time_it([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0); }, R);
    printf("  %-34s %8.2f us\n", "empty kernel 1024 WGs", us);
    for (int grid : {512, 1024, 2048, 4096}) {
      us = time_it([&] { hipLaunchKernelGGL((k_diag<0>), dim3(grid), dim3(256), 0, 0, z, nc, d.col, d.val, d.x, d.y); }, R);
      printf("  diag stream-only grid=%-5d          %8.2f us  %7.1f GB/s\n", grid, us, 12.0 * z / us / 1e3);
    }
    us = time_it([&] { hipLaunchKernelGGL((k_diag<1>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.val, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "diag stream + coalesced x", us);
    us = time_it([&] { hipLaunchKernelGGL((k_diag<2>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.val, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "diag stream + gather in 32KB", us);
    us = time_it([&] { hipLaunchKernelGGL((k_diag<3>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.val, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "diag stream + full gather", us);
    us = time_it([&] { hipLaunchKernelGGL(k_gather_only, dim3(2048), dim3(256), 0, 0, z, nc, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "diag gather only (hashed idx)", us);
    us = time_it([&] { hipLaunchKernelGGL((k_gather_pol<0>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "gather(col) plain", us);
    us = time_it([&] { hipLaunchKernelGGL((k_gather_pol<1>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "gather(col) nontemporal", us);
    us = time_it([&] { hipLaunchKernelGGL((k_gather_pol<2>), dim3(2048), dim3(256), 0, 0, z, nc, d.col, d.x, d.y); }, R);
    printf("  %-34s %8.2f us\n", "gather(col) sc1", us);
    for (int grid : {1024, 4096, 8192}) {
      us = time_it([&] { hipLaunchKernelGGL((k_gather_pol<0>), dim3(grid), dim3(256), 0, 0, z, nc, d.col, d.x, d.y); }, R);
      printf("  gather(col) plain grid=%-5d         %8.2f us\n", grid, us);
    }
  }
#define STREAM(BS, T, NT, GRIDCAP)                                                                                     \
  {                                                                                                                    \
    build_rb(M, T, 4 * BS, rb); d.nb = (int)rb.size() - 1;                                                             \
    CK(hipMemcpy(d.rb, rb.data(), sizeof(int) * rb.size(), hipMemcpyHostToDevice));                                    \
    int grid = std::min(d.nb, GRIDCAP);                                                                                \
    CK(hipMemset(d.y, 0, sizeof(double) * nr));                                                                        \
    double us = time_it([&] { hipLaunchKernelGGL((k_stream<BS, T, NT>), dim3(grid), dim3(BS), 0, 0, d.rp, d.col, d.val, d.rb, d.nb, d.x, d.y); }, R); \
    char nm[96]; snprintf(nm, 96, "stream BS=%d T=%d nt=%d grid=%d", BS, T, (int)NT, grid); check(nm, M, x, d.y, us, bytes);  \
  }
  STREAM(256, 4096, false, 1024)
  STREAM(256, 2048, false, 4096)
  STREAM(256, 1024, false, 8192)
  STREAM(256, 1024, true, 8192)
  STREAM(256, 512, false, 16384)
  STREAM(128, 512, false, 16384)
  STREAM(512, 4096, false, 4096)
  STREAM(512, 2048, false, 4096)
  STREAM(1024, 4096, false, 4096)
  STREAM(256, 2048, false, 512)
  STREAM(256, 1024, false, 1024)
  STREAM(256, 1024, false, 2048)
#define SORTED(BS, T)                                                                                                  \
  {                                                                                                                    \
    build_rb(M, T, 8 * BS, rb); d.nb = (int)rb.size() - 1;                                                             \
    std::vector<int> scol(z); std::vector<double> sval(z); std::vector<unsigned short> spos(z);                        \
    for (int t = 0; t < d.nb; ++t) {                                                                                   \
      const int a = M.rp[rb[t]], b = M.rp[rb[t + 1]];                                                                  \
      std::vector<int> idx(b - a); for (int i = 0; i < b - a; ++i) idx[i] = i;                                         \
      std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return M.col[a + p] < M.col[a + q]; });             \
      for (int i = 0; i < b - a; ++i) { scol[a + i] = M.col[a + idx[i]]; sval[a + i] = M.val[a + idx[i]]; spos[a + i] = (unsigned short)idx[i]; } \
    }                                                                                                                  \
    int* dcol; double* dval; unsigned short* dpos;                                                                     \
    CK(hipMalloc(&dcol, sizeof(int) * z)); CK(hipMalloc(&dval, sizeof(double) * z)); CK(hipMalloc(&dpos, 2 * z));      \
    CK(hipMemcpy(dcol, scol.data(), sizeof(int) * z, hipMemcpyHostToDevice));                                          \
    CK(hipMemcpy(dval, sval.data(), sizeof(double) * z, hipMemcpyHostToDevice));                                       \
    CK(hipMemcpy(dpos, spos.data(), 2 * z, hipMemcpyHostToDevice));                                                    \
    CK(hipMemcpy(d.rb, rb.data(), sizeof(int) * rb.size(), hipMemcpyHostToDevice));                                    \
    CK(hipMemset(d.y, 0, sizeof(double) * nr));                                                                        \
    double us = time_it([&] { hipLaunchKernelGGL((k_stream_sorted<BS, T>), dim3(d.nb), dim3(BS), 0, 0, d.rp, dcol, dval, dpos, d.rb, d.nb, d.x, d.y); }, R); \
    char nm[96]; snprintf(nm, 96, "SORTED BS=%d T=%d grid=%d", BS, T, d.nb); check(nm, M, x, d.y, us, bytes);          \
    CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(dpos));                                                           \
  }
  SORTED(256, 2048)
  SORTED(256, 4096)
  SORTED(512, 4096)
  SORTED(512, 8192)
  SORTED(1024, 8192)
  SORTED(1024, 16384)
#define PIPE(BS, T, NT)                                                                                                \
  {                                                                                                                    \
    build_rb(M, T, 4 * BS, rb); d.nb = (int)rb.size() - 1;                                                             \
    CK(hipMemcpy(d.rb, rb.data(), sizeof(int) * rb.size(), hipMemcpyHostToDevice));                                    \
    int grid = (d.nb + NT - 1) / NT;                                                                                   \
    CK(hipMemset(d.y, 0, sizeof(double) * nr));                                                                        \
    double us = time_it([&] { hipLaunchKernelGGL((k_stream_pipe<BS, T, NT>), dim3(grid), dim3(BS), 0, 0, d.rp, d.col, d.val, d.rb, d.nb, d.x, d.y); }, R); \
    char nm[96]; snprintf(nm, 96, "PIPE BS=%d T=%d NT=%d grid=%d", BS, T, NT, grid); check(nm, M, x, d.y, us, bytes);  \
  }
  PIPE(256, 2048, 2)
  PIPE(256, 2048, 4)
  PIPE(256, 1024, 2)
  PIPE(256, 1024, 4)
  PIPE(256, 1024, 8)
  PIPE(256, 512, 4)
  PIPE(256, 512, 8)
  PIPE(512, 2048, 2)
  PIPE(512, 2048, 4)
#define STREAMG(BS, T, G)                                                                                              \
  {                                                                                                                    \
    build_rb(M, T, 4 * BS, rb); d.nb = (int)rb.size() - 1;                                                             \
    CK(hipMemcpy(d.rb, rb.data(), sizeof(int) * rb.size(), hipMemcpyHostToDevice));                                    \
    CK(hipMemset(d.y, 0, sizeof(double) * nr));                                                                        \
    double us = time_it([&] { hipLaunchKernelGGL((k_stream_g<BS, T, G>), dim3(d.nb), dim3(BS), 0, 0, d.rp, d.col, d.val, d.rb, d.nb, d.x, d.y); }, R); \
    char nm[96]; snprintf(nm, 96, "stream_g BS=%d T=%d G=%d", BS, T, G); check(nm, M, x, d.y, us, bytes);              \
  }
  STREAMG(256, 2048, 2)
  STREAMG(256, 2048, 4)
  STREAMG(256, 1024, 4)
  STREAMG(256, 4096, 4)
#define VECTOR(BS, L, NT, WPR)                                                                                         \
  {                                                                                                                    \
    long long thr = (long long)nr * L; int grid = (int)std::min<long long>((thr + BS - 1) / BS, (long long)WPR);       \
    CK(hipMemset(d.y, 0, sizeof(double) * nr));                                                                        \
    double us = time_it([&] { hipLaunchKernelGGL((k_vector<BS, L, NT>), dim3(grid), dim3(BS), 0, 0, nr, d.rp, d.col, d.val, d.x, d.y); }, R); \
    char nm[96]; snprintf(nm, 96, "vector BS=%d L=%d nt=%d grid=%d", BS, L, (int)NT, grid); check(nm, M, x, d.y, us, bytes); \
  }
  VECTOR(256, 2, false, 1 << 20)
  VECTOR(256, 4, false, 1 << 20)
  VECTOR(256, 8, false, 1 << 20)
  VECTOR(256, 16, false, 1 << 20)
  VECTOR(256, 4, true, 1 << 20)
  VECTOR(256, 8, true, 1 << 20)
  VECTOR(256, 4, false, 2048)
  VECTOR(256, 8, false, 2048)
  VECTOR(256, 8, false, 4096)
  VECTOR(512, 8, false, 2048)
  // plain copy roofline reference: read val+col, write y-sized
  CK(hipFree(d.rp)); CK(hipFree(d.col)); CK(hipFree(d.val)); CK(hipFree(d.x)); CK(hipFree(d.y)); CK(hipFree(d.rb));
}

int main(int argc, char** argv) {
  int which = argc > 1 ? atoi(argv[1]) : 0;
  if (which == 0 || which == 1) run_matrix("A   ", 200000, 100000, 2000000, 1);
  if (which == 0 || which == 2) run_matrix("A^T ", 100000, 200000, 2000000, 2);
  if (which == 0 || which == 3) run_matrix("[P|A^T]", 100000, 300000, 2500000, 3);
  return 0;
}
