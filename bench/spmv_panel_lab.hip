// spmv_panel_lab.hip -- laboratory for the column-panel ("2-D blocked") SpMV: the gathered vector is staged through LDS one
// panel of W columns at a time, so the random 8-byte gathers hit LDS instead of L2 (bench/spmv_lab.hip measured the L2 gather
// rate as the bound of the CSR-stream kernel).  A (panel, row-chunk) tile writes partial row sums; a second kernel folds the
// panels in fixed order.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off bench/spmv_panel_lab.hip -o bench/spmv_panel_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Csr { int nr, nc; std::vector<int> rp, col; std::vector<double> val; };

static Csr random_csr(int nr, int nc, long long nnz, unsigned seed) {
  std::mt19937_64 g(seed);
  std::vector<std::pair<long long, double>> e(nnz);
  std::normal_distribution<double> nd;
  for (auto& p : e) { p.first = (long long)(g() % nr) * nc + (long long)(g() % nc); p.second = nd(g); }
  std::sort(e.begin(), e.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  Csr M; M.nr = nr; M.nc = nc; M.rp.assign(nr + 1, 0);
  long long last = -1;
  for (auto& p : e) {
    if (p.first == last) { M.val.back() += p.second; continue; }
    last = p.first; M.col.push_back((int)(p.first % nc)); M.val.push_back(p.second); M.rp[p.first / nc + 1]++;
  }
  for (int i = 0; i < nr; ++i) M.rp[i + 1] += M.rp[i];
  return M;
}

struct Panels {
  int np, W, nr;
  std::vector<int> prp;             // np * (nr + 1): absolute offsets into pcol / pval
  std::vector<unsigned short> pcol; // column - p * W
  std::vector<double> pval;
  std::vector<int> tiles;           // 4 ints per tile: {p, r0, r1, 0}
};

static Panels build_panels(const Csr& M, int W, int tile_nnz, int rows_max) {
  Panels P; P.W = W; P.nr = M.nr; P.np = (M.nc + W - 1) / W;
  const int nr = M.nr;
  P.prp.assign((size_t)P.np * (nr + 1), 0);
  std::vector<long long> cnt(P.np, 0);
  for (int r = 0; r < nr; ++r)
    for (int k = M.rp[r]; k < M.rp[r + 1]; ++k) { int p = M.col[k] / W; P.prp[(size_t)p * (nr + 1) + r + 1]++; cnt[p]++; }
  long long off = 0;
  for (int p = 0; p < P.np; ++p) {
    int* rp = &P.prp[(size_t)p * (nr + 1)];
    rp[0] = (int)off;
    for (int r = 0; r < nr; ++r) rp[r + 1] += rp[r];
    off += cnt[p];
  }
  P.pcol.resize(off); P.pval.resize(off);
  std::vector<int> fill(P.prp);
  for (int r = 0; r < nr; ++r)
    for (int k = M.rp[r]; k < M.rp[r + 1]; ++k) {
      int p = M.col[k] / W; int& f = fill[(size_t)p * (nr + 1) + r];
      P.pcol[f] = (unsigned short)(M.col[k] - p * W); P.pval[f] = M.val[k]; ++f;
    }
  for (int p = 0; p < P.np; ++p) {
    const int* rp = &P.prp[(size_t)p * (nr + 1)];
    int r = 0;
    while (r < nr) {
      int r1 = r;
      while (r1 < nr && r1 - r < rows_max && rp[r1 + 1] - rp[r] <= tile_nnz) ++r1;
      if (r1 == r) r1 = r + 1;
      P.tiles.push_back(p); P.tiles.push_back(r); P.tiles.push_back(r1); P.tiles.push_back(0);
      r = r1;
    }
  }
  return P;
}

// V1: one thread per row of the tile, gathers from the LDS panel.
template <int BS, int W>
__global__ __launch_bounds__(BS) void k_panel_rows(const int* __restrict__ prp, const unsigned short* __restrict__ pcol,
                                                   const double* __restrict__ pval, const int4* __restrict__ tiles,
                                                   const double* __restrict__ x, int nc, int nr, double* __restrict__ part) {
  extern __shared__ double xs[];
  const int4 t = tiles[blockIdx.x];
  const int p = t.x, r0 = t.y, r1 = t.z;
  const int c0 = p * W;
  const int cw = min(W, nc - c0);
  const int* rp = prp + (size_t)p * (nr + 1);
  for (int i = threadIdx.x * 2; i < cw; i += BS * 2) {
    if (i + 1 < cw) { const double2 v = *reinterpret_cast<const double2*>(x + c0 + i); xs[i] = v.x; xs[i + 1] = v.y; }
    else xs[i] = x[c0 + i];
  }
  __syncthreads();
  double* out = part + (size_t)p * nr;
  for (int r = r0 + threadIdx.x; r < r1; r += BS) {
    const int a = rp[r], b = rp[r + 1];
    double s = 0.0;
    for (int k = a; k < b; ++k) s += pval[k] * xs[pcol[k]];
    out[r] = s;
  }
}

// V2: coalesced (val, col) stream -> products in an LDS chunk -> one thread per row sums its segment (CSR-stream inside the tile).
template <int BS, int W, int CH>
__global__ __launch_bounds__(BS) void k_panel_stream(const int* __restrict__ prp, const unsigned short* __restrict__ pcol,
                                                     const double* __restrict__ pval, const int4* __restrict__ tiles,
                                                     const double* __restrict__ x, int nc, int nr, double* __restrict__ part) {
  extern __shared__ double xs[];
  double* prod = xs + W;
  const int4 t = tiles[blockIdx.x];
  const int p = t.x, r0 = t.y, r1 = t.z;
  const int c0 = p * W;
  const int cw = min(W, nc - c0);
  const int* rp = prp + (size_t)p * (nr + 1);
  const int nz0 = rp[r0], nz1 = rp[r1];
  // issue the matrix loads of the first chunk before staging x so both are in flight together
  double v[CH / BS]; unsigned short c[CH / BS];
#pragma unroll
  for (int j = 0; j < CH / BS; ++j) { const int k = nz0 + j * BS + threadIdx.x; if (k < nz1) { v[j] = pval[k]; c[j] = pcol[k]; } }
  for (int i = threadIdx.x * 2; i < cw; i += BS * 2) {
    if (i + 1 < cw) { const double2 w = *reinterpret_cast<const double2*>(x + c0 + i); xs[i] = w.x; xs[i + 1] = w.y; }
    else xs[i] = x[c0 + i];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CH / BS; ++j) { const int k = j * BS + threadIdx.x; if (nz0 + k < nz1) prod[k] = v[j] * xs[c[j]]; }
  __syncthreads();
  double* out = part + (size_t)p * nr;
  for (int r = r0 + threadIdx.x; r < r1; r += BS) {
    const int a = rp[r] - nz0, b = rp[r + 1] - nz0;
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    out[r] = s;
  }
}


// V3: persistent workgroup = one panel's x in LDS for its whole life + a software pipeline over row tiles: the (val, col, rowptr)
// loads of tile t+1 are in flight while tile t's products are summed; products are double-buffered so one barrier per tile.
template <int BS, int W, int CH, int MAXT>
__global__ __launch_bounds__(BS) void k_panel_persist(const int* __restrict__ prp, const unsigned short* __restrict__ pcol,
                                                      const double* __restrict__ pval, const int4* __restrict__ tdesc,
                                                      const int4* __restrict__ wdesc, const double* __restrict__ x, int nc, int nr,
                                                      double* __restrict__ part) {
  extern __shared__ double xs[];
  double* prod = xs + W;                       // 2 * CH
  int4* tds = reinterpret_cast<int4*>(prod + 2 * CH);
  const int4 wd = wdesc[2 * blockIdx.x];
  int4 d = wdesc[2 * blockIdx.x + 1];          // first tile {r0, r1, nz0, nz1}
  const int p = wd.x, t0 = wd.y, nt = wd.z - wd.y;
  const int* rp = prp + (size_t)p * (nr + 1);
  double v[CH / BS]; unsigned short c[CH / BS]; int ra = 0, rb = 0;
#define LOAD_TILE(D)                                                                                                   \
  {                                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < CH / BS; ++j) {                                                              \
      const int k = (D).z + j * BS + (int)threadIdx.x;                                                                 \
      if (k < (D).w) { v[j] = pval[k]; c[j] = pcol[k]; }                                                               \
    }                                                                                                                  \
    const int r = (D).x + (int)threadIdx.x;                                                                            \
    if (r < (D).y) { ra = rp[r]; rb = rp[r + 1]; }                                                                     \
  }
  LOAD_TILE(d)
  if ((int)threadIdx.x < nt) tds[threadIdx.x] = tdesc[t0 + threadIdx.x];
  const int c0 = p * W;
  const int cw = min(W, nc - c0);
  for (int i = threadIdx.x * 2; i < cw; i += BS * 2) {
    if (i + 1 < cw) { const double2 w = *reinterpret_cast<const double2*>(x + c0 + i); xs[i] = w.x; xs[i + 1] = w.y; }
    else xs[i] = x[c0 + i];
  }
  __syncthreads();
  double* out = part + (size_t)p * nr;
  for (int t = 0; t < nt; ++t) {
    const int4 cur = d;
    double* buf = prod + (t & 1) * CH;
#pragma unroll
    for (int j = 0; j < CH / BS; ++j) { const int k = j * BS + threadIdx.x; if (cur.z + k < cur.w) buf[k] = v[j] * xs[c[j]]; }
    const int a = ra - cur.z, b = rb - cur.z;
    const int myrow = cur.x + threadIdx.x;
    if (t + 1 < nt) { d = tds[t + 1]; LOAD_TILE(d) }
    __syncthreads();
    if (myrow < cur.y) {
      double s = 0.0;
      for (int k = a; k < b; ++k) s += buf[k];
      out[myrow] = s;
    }
  }
#undef LOAD_TILE
}

template <int BS>
__global__ __launch_bounds__(BS) void k_combine(const double* __restrict__ part, int np, int nr, double* __restrict__ y) {
  const int r = blockIdx.x * BS + threadIdx.x;
  if (r >= nr) return;
  double s = 0.0;
  for (int p = 0; p < np; ++p) s += part[(size_t)p * nr + r];
  y[r] = s;
}

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

static double maxerr_vs_cpu(const Csr& M, const std::vector<double>& x, const double* dy) {
  std::vector<double> y(M.nr);
  CK(hipMemcpy(y.data(), dy, sizeof(double) * M.nr, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int i = 0; i < M.nr; ++i) {
    double s = 0; for (int j = M.rp[i]; j < M.rp[i + 1]; ++j) s += M.val[j] * x[M.col[j]];
    maxerr = fmax(maxerr, fabs(s - y[i]));
  }
  return maxerr;
}

template <int BS, int W, int VAR, int CH>
static void run_variant(const Csr& M, const std::vector<double>& x, const double* dx, double* dy, int tile_nnz, int rows_max, double bytes) {
  Panels P = build_panels(M, W, VAR == 2 ? CH : tile_nnz, rows_max);
  const int nt = (int)P.tiles.size() / 4;
  int *dprp, *dtiles; unsigned short* dcol; double *dval, *dpart;
  CK(hipMalloc(&dprp, sizeof(int) * P.prp.size())); CK(hipMalloc(&dtiles, sizeof(int) * P.tiles.size()));
  CK(hipMalloc(&dcol, 2 * P.pcol.size())); CK(hipMalloc(&dval, 8 * P.pval.size())); CK(hipMalloc(&dpart, 8 * (size_t)P.np * M.nr));
  CK(hipMemcpy(dprp, P.prp.data(), sizeof(int) * P.prp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtiles, P.tiles.data(), sizeof(int) * P.tiles.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, P.pcol.data(), 2 * P.pcol.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dval, P.pval.data(), 8 * P.pval.size(), hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0, 8 * (size_t)M.nr));
  const size_t lds = (size_t)W * 8 + (VAR == 2 ? (size_t)CH * 8 : 0);
  auto main_k = [&] {
    if constexpr (VAR == 1) hipLaunchKernelGGL((k_panel_rows<BS, W>), dim3(nt), dim3(BS), lds, 0, dprp, dcol, dval, (const int4*)dtiles, dx, M.nc, M.nr, dpart);
    else hipLaunchKernelGGL((k_panel_stream<BS, W, CH>), dim3(nt), dim3(BS), lds, 0, dprp, dcol, dval, (const int4*)dtiles, dx, M.nc, M.nr, dpart);
  };
  auto comb_k = [&] { hipLaunchKernelGGL((k_combine<256>), dim3((M.nr + 255) / 256), dim3(256), 0, 0, dpart, P.np, M.nr, dy); };
  if constexpr (VAR == 1) CK(hipFuncSetAttribute((const void*)k_panel_rows<BS, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else CK(hipFuncSetAttribute((const void*)k_panel_stream<BS, W, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int R = 200;
  const double t_main = time_it(main_k, R);
  const double t_comb = time_it(comb_k, R);
  const double t_both = time_it([&] { main_k(); comb_k(); }, R);
  const double err = maxerr_vs_cpu(M, x, dy);
  printf("  panel V%d BS=%-4d W=%-5d tile=%-5d np=%-2d tiles=%-5d main %6.2f  combine %5.2f  both %6.2f us  %7.1f GB/s  maxerr %.1e\n", VAR, BS, W,
         VAR == 2 ? CH : tile_nnz, P.np, nt, t_main, t_comb, t_both, bytes / t_both / 1e3, err);
  CK(hipFree(dprp)); CK(hipFree(dtiles)); CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(dpart));
}


template <int BS, int W, int CH, int MAXT>
static void run_persist(const Csr& M, const std::vector<double>& x, const double* dx, double* dy, int target_wgs, double bytes) {
  Panels P = build_panels(M, W, CH, BS);
  const int nt = (int)P.tiles.size() / 4;
  // tile descriptors {r0, r1, nz0, nz1} and workgroup descriptors {p, t0, t1, 0}{first tile}
  std::vector<int> td(4 * (size_t)nt), wd;
  std::vector<int> pstart(P.np + 1, 0);
  for (int t = 0; t < nt; ++t) {
    const int p = P.tiles[4 * t], r0 = P.tiles[4 * t + 1], r1 = P.tiles[4 * t + 2];
    const int* rp = &P.prp[(size_t)p * (M.nr + 1)];
    td[4 * t] = r0; td[4 * t + 1] = r1; td[4 * t + 2] = rp[r0]; td[4 * t + 3] = rp[r1];
    pstart[p + 1] = t + 1;
  }
  for (int p = 0; p < P.np; ++p) if (pstart[p + 1] == 0) pstart[p + 1] = pstart[p];
  int nwg = 0;
  for (int p = 0; p < P.np; ++p) {
    const int a = pstart[p], b = pstart[p + 1], cnt = b - a;
    if (cnt == 0) continue;
    int g = std::max(1, (int)llround((double)target_wgs * cnt / nt));
    while ((cnt + g - 1) / g > MAXT) ++g;
    for (int i = 0; i < g; ++i) {
      const int t0 = a + (int)((long long)cnt * i / g), t1 = a + (int)((long long)cnt * (i + 1) / g);
      if (t1 == t0) continue;
      wd.insert(wd.end(), {p, t0, t1, 0, td[4 * t0], td[4 * t0 + 1], td[4 * t0 + 2], td[4 * t0 + 3]});
      ++nwg;
    }
  }
  int *dprp, *dtd, *dwd; unsigned short* dcol; double *dval, *dpart;
  CK(hipMalloc(&dprp, sizeof(int) * P.prp.size())); CK(hipMalloc(&dtd, sizeof(int) * td.size())); CK(hipMalloc(&dwd, sizeof(int) * wd.size()));
  CK(hipMalloc(&dcol, 2 * P.pcol.size())); CK(hipMalloc(&dval, 8 * P.pval.size())); CK(hipMalloc(&dpart, 8 * (size_t)P.np * M.nr));
  CK(hipMemcpy(dprp, P.prp.data(), sizeof(int) * P.prp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtd, td.data(), sizeof(int) * td.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dwd, wd.data(), sizeof(int) * wd.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, P.pcol.data(), 2 * P.pcol.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dval, P.pval.data(), 8 * P.pval.size(), hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0, 8 * (size_t)M.nr));
  const size_t lds = (size_t)W * 8 + 2 * (size_t)CH * 8 + 16 * MAXT;
  CK(hipFuncSetAttribute((const void*)k_panel_persist<BS, W, CH, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto main_k = [&] { hipLaunchKernelGGL((k_panel_persist<BS, W, CH, MAXT>), dim3(nwg), dim3(BS), lds, 0, dprp, dcol, dval, (const int4*)dtd, (const int4*)dwd, dx, M.nc, M.nr, dpart); };
  auto comb_k = [&] { hipLaunchKernelGGL((k_combine<256>), dim3((M.nr + 255) / 256), dim3(256), 0, 0, dpart, P.np, M.nr, dy); };
  const int R = 200;
  const double t_main = time_it(main_k, R);
  const double t_comb = time_it(comb_k, R);
  const double t_both = time_it([&] { main_k(); comb_k(); }, R);
  const double err = maxerr_vs_cpu(M, x, dy);
  printf("  panel V3 BS=%-4d W=%-5d CH=%-5d np=%-2d tiles=%-5d wgs=%-4d main %6.2f  combine %5.2f  both %6.2f us  %7.1f GB/s  maxerr %.1e\n", BS, W, CH,
         P.np, nt, nwg, t_main, t_comb, t_both, bytes / t_both / 1e3, err);
  CK(hipFree(dprp)); CK(hipFree(dtd)); CK(hipFree(dwd)); CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(dpart));
}

static void run_matrix(const char* label, int nr, int nc, long long nnz, unsigned seed) {
  Csr M = random_csr(nr, nc, nnz, seed);
  const long long z = (long long)M.val.size();
  const double bytes = 12.0 * z + 4.0 * (nr + 1) + 8.0 * nc + 8.0 * nr;
  printf("%s: %d x %d, nnz %lld (%.1f/row), algorithmic bytes %.2f MB, roofline@8TB/s %.2f us\n", label, nr, nc, z, (double)z / nr,
         bytes / 1e6, bytes / 8e6);
  std::vector<double> x(nc); std::mt19937_64 g(7); std::normal_distribution<double> nd; for (auto& v : x) v = nd(g);
  double *dx, *dy;
  CK(hipMalloc(&dx, 8 * (size_t)nc)); CK(hipMalloc(&dy, 8 * (size_t)nr));
  CK(hipMemcpy(dx, x.data(), 8 * (size_t)nc, hipMemcpyHostToDevice));
  run_persist<1024, 15360, 2048, 32>(M, x, dx, dy, 256, bytes);
  run_persist<1024, 15360, 2048, 32>(M, x, dx, dy, 512, bytes);
  run_persist<1024, 15360, 2048, 32>(M, x, dx, dy, 128, bytes);
  run_persist<1024, 16384, 1024, 64>(M, x, dx, dy, 256, bytes);
  run_persist<512, 8192, 1024, 64>(M, x, dx, dy, 512, bytes);
  run_persist<1024, 8192, 2048, 32>(M, x, dx, dy, 512, bytes);
  run_persist<256, 8192, 512, 128>(M, x, dx, dy, 512, bytes);
  run_persist<256, 4096, 512, 128>(M, x, dx, dy, 1024, bytes);
  run_variant<1024, 16384, 1, 0>(M, x, dx, dy, 8192, 1 << 20, bytes);
  run_variant<1024, 16384, 1, 0>(M, x, dx, dy, 4096, 1 << 20, bytes);
  run_variant<1024, 16384, 1, 0>(M, x, dx, dy, 16384, 1 << 20, bytes);
  run_variant<512, 16384, 1, 0>(M, x, dx, dy, 8192, 1 << 20, bytes);
  run_variant<512, 8192, 1, 0>(M, x, dx, dy, 4096, 1 << 20, bytes);
  run_variant<512, 8192, 1, 0>(M, x, dx, dy, 8192, 1 << 20, bytes);
  run_variant<256, 8192, 1, 0>(M, x, dx, dy, 4096, 1 << 20, bytes);
  run_variant<256, 4096, 1, 0>(M, x, dx, dy, 2048, 1 << 20, bytes);
  run_variant<1024, 16384, 2, 4096>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<1024, 16384, 2, 2048>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<512, 16384, 2, 4096>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<512, 8192, 2, 4096>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<512, 8192, 2, 2048>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<256, 8192, 2, 2048>(M, x, dx, dy, 0, 1 << 20, bytes);
  run_variant<256, 4096, 2, 2048>(M, x, dx, dy, 0, 1 << 20, bytes);
  CK(hipFree(dx)); CK(hipFree(dy));
}

int main(int argc, char** argv) {
  int which = argc > 1 ? atoi(argv[1]) : 0;
  if (which == 0 || which == 1) run_matrix("A   ", 200000, 100000, 2000000, 1);
  if (which == 0 || which == 3) run_matrix("[P|A^T]", 100000, 300000, 2500000, 3);
  return 0;
}
