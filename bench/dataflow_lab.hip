// dataflow_lab.hip -- lab for VERDICT r05 item 2: per-cone dependency counters inside ONE persistent launch instead of one launch per product.
// The scheduling skeleton of csrc/psd_polar.hip: k_polar_dataflow with a synthetic tile body, on a synthetic batch shaped like BASELINE config 5
// (400 cones, 1-10 tiles each, pinned to XCDs by list scheduling, 44 products):
//   * every XCD has one in-order queue (item q = tile q mod n_x of product q div n_x), persistent workgroups take items by ticket;
//   * a tile of product p waits until done[cone] >= p * tiles(cone); a finished tile: plain stores -> s_waitcnt vmcnt(0) -> barrier -> done[cone] += 1; the consumer
//     reads the operands with sc1 loads (L1 bypass, served by the XCD's L2 -- the coherence point of producer and consumer, which sit in the same XCD by
//     construction).  First version of this lab used plain loads behind `buffer_inv sc0`: WRONG (sc0 is a workgroup-scope invalidate with no effect on
//     the L1; stale reads as soon as a CU re-reads a small region it has cached -- variant `inv0` below keeps that form to show it);
//   * tile body: read the cone's whole region (must hold the value p everywhere), spin `work` clocks, write the tile's slice with p + 1.
// Reports: wall time of the persistent launch vs. 44 launches of the same tiles, stale reads, timeouts, workgroups per XCD.
//   hipcc --offload-arch=gfx950 -O3 bench/dataflow_lab.hip -o bench/dataflow_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdlib.h>
#include <vector>
#include <algorithm>

struct Tile { int cone, slice, nslices, pad; long long off; int elems, work; };
#define SYNC_DONE 144
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7u; }
__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void tile_body(const Tile& td, double* buf, int p, unsigned long long* stale) {
  double* reg = buf + td.off;
  double acc = 0.0;
  for (int i = threadIdx.x; i < td.elems; i += 256) { const double v = reg[i]; acc += (v - p) * (v - p); }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(stale, 1ull);
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)td.work) {}
  __syncthreads();                                     // every reader of this workgroup is done before the slice is overwritten (other tiles of the same product only READ other slices' old value? no: they
                                                       // read the WHOLE region, so a slice may only be written when all tiles of product p have read it -- the real kernel writes a DIFFERENT buffer; here
                                                       // the value check therefore accepts p or p + 1)
  const int per = td.elems / td.nslices;
  for (int i = threadIdx.x; i < per; i += 256) reg[td.slice * per + i] = (double)(p + 1);
}
template <int SC1>
__device__ __forceinline__ void tile_body2(const Tile& td, double* buf, int p, unsigned long long* stale) {
  // two buffers like the real iteration (read buffer p & 1, write buffer (p + 1) & 1): the check is exact
  double* rd = buf + 2 * td.off + (size_t)(p & 1) * td.elems;
  double* wr = buf + 2 * td.off + (size_t)((p + 1) & 1) * td.elems;
  double acc = 0.0;
  for (int i = threadIdx.x; i < td.elems; i += 256) { const double v = SC1 ? __hip_atomic_load(rd + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rd[i]; acc += (v - p) * (v - p); }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(stale, 1ull);
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)td.work) {}
  const int per = td.elems / td.nslices;
  for (int i = threadIdx.x; i < per; i += 256) wr[td.slice * per + i] = (double)(p + 1);
}

template <int SC1>
__global__ __launch_bounds__(256) void k_dataflow(const Tile* tiles, const int* cone_nt, unsigned* sync, double* buf, int nprod, const int* xoff, unsigned long long* stale,
                                                  unsigned* wg_per_xcd) {
  __shared__ unsigned s_item;
  __shared__ int s_fail;
  const int x = (int)xcc_id();
  const int x0 = xoff[x];
  const unsigned n_x = (unsigned)(xoff[x + 1] - x0), total = n_x * (unsigned)nprod;
  __shared__ int s_ready;
  const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;     // SCALAR branch condition (see the note in the loop)
  if (wave0) { if (threadIdx.x == 0) { s_fail = 0; s_ready = 0; atomicAdd(wg_per_xcd + x, 1u); s_item = n_x ? __hip_atomic_fetch_add(sync + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; } }
  __syncthreads();
  int prev_cone = -1;
  unsigned nxt = 0;
  for (;;) {
    // Per tile: ONE barrier (two when the tile has to wait).  Everything on thread 0's dependent chain -- the ticket of the NEXT item, that item's cone, the first look
    // at the cone's completion counter -- is requested while the current tile runs or while the other waves drain their stores; every loop exit and every branch
    // around a barrier is decided on SCALAR values (a per-lane condition, or thread-0 regions on both sides of the back edge without a barrier between them,
    // make the compiler build an exec-masked loop nest in which the waves' barrier counts diverge: a hang, reproduced by the first version of this lab).
    const unsigned item = __builtin_amdgcn_readfirstlane(s_item);
    const int ready = __builtin_amdgcn_readfirstlane(s_ready);
    const bool more = item < total;
    unsigned p = 0, t = 0;
    if (more) { p = item / n_x; t = item - p * n_x; }
    const Tile td = tiles[x0 + t];
    if (wave0) {
      if (threadIdx.x == 0) {
        if (prev_cone >= 0) __hip_atomic_fetch_add(sync + SYNC_DONE + prev_cone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the previous tile: complete since the barrier at the loop's tail
        if (more && !ready) {
          const unsigned target = p * (unsigned)cone_nt[td.cone];
          long sp = 0;
          while (ldu(sync + SYNC_DONE + td.cone) < target) {
            __builtin_amdgcn_s_sleep(1);
            ++sp;
            if ((sp & 255) == 0 && ldu(sync + 128)) { s_fail = 1; break; }
            if (sp > (1L << 18)) { s_fail = 1; __hip_atomic_store(sync + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(sync + 129, 1u); break; }
          }
        }
      }
    }
    if (!more) break;
    if (!ready) {
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_fail)) break;
    }
    if (wave0) { if (threadIdx.x == 0) nxt = __hip_atomic_fetch_add(sync + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }    // the NEXT ticket: in flight during the tile
    tile_body2<SC1>(td, buf, (int)p, stale);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave0) {
      if (threadIdx.x == 0) {        // while the other waves drain their stores: the next item and a first look at its dependency
        s_item = nxt;
        int rdy = 1;
        if (nxt < total) {
          const unsigned p2 = nxt / n_x, t2 = nxt - p2 * n_x;
          if (p2 > 0) { const int c2 = tiles[x0 + t2].cone; rdy = ldu(sync + SYNC_DONE + c2) >= p2 * (unsigned)cone_nt[c2]; }
        }
        s_ready = rdy;
      }
    }
    __syncthreads();
    prev_cone = td.cone;
  }
}

// launch-per-product form: workgroup b takes tile b of the XCD-interleaved list
__global__ __launch_bounds__(256) void k_product(const Tile* tiles_il, double* buf, int p, unsigned long long* stale) {
  const Tile td = tiles_il[blockIdx.x];
  if (td.cone < 0) return;
  tile_body2<0>(td, buf, p, stale);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int nprod = argc > 1 ? atoi(argv[1]) : 44;
  const int work = argc > 2 ? atoi(argv[2]) : 25000;       // ~ clocks of a tile body (a real tile lives ~30 k shader clocks)
  const int grid = argc > 3 ? atoi(argv[3]) : 1024;
  const int ncones = argc > 4 ? atoi(argv[4]) : 400;
  const int sc1 = argc > 5 ? atoi(argv[5]) : 1;
  srand(5);
  std::vector<int> nt(ncones), d(ncones);
  for (int c = 0; c < ncones; ++c) { d[c] = 20 + rand() % 181; const int t = (d[c] + 63) / 64; nt[c] = t * (t + 1) / 2; }
  std::vector<int> order(ncones); for (int c = 0; c < ncones; ++c) order[c] = c;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] > d[b]; });
  std::vector<std::vector<Tile>> xl(8);
  long long load[8] = {0}; long long off = 0;
  std::vector<long long> coff(ncones);
  for (int c = 0; c < ncones; ++c) { coff[c] = off; const int elems = ((d[c] * d[c] + nt[c] * 256 - 1) / (nt[c] * 256)) * nt[c] * 256; off += elems; }
  for (int c : order) {
    int x = 0; for (int t = 1; t < 8; ++t) if (load[t] < load[x]) x = t;
    const int elems = ((d[c] * d[c] + nt[c] * 256 - 1) / (nt[c] * 256)) * nt[c] * 256;
    for (int s = 0; s < nt[c]; ++s) { Tile t; t.cone = c; t.slice = s; t.nslices = nt[c]; t.pad = 0; t.off = coff[c]; t.elems = elems; t.work = work * std::max(d[c], 64) / 200; xl[x].push_back(t); }
    load[x] += (long long)nt[c] * d[c];
  }
  for (int x = 0; x < 8; ++x) std::stable_sort(xl[x].begin(), xl[x].end(), [](const Tile& a, const Tile& b) { return a.work > b.work; });
  std::vector<Tile> dl; std::vector<int> xoff(9);
  for (int x = 0; x < 8; ++x) { xoff[x] = (int)dl.size(); dl.insert(dl.end(), xl[x].begin(), xl[x].end()); }
  xoff[8] = (int)dl.size();
  size_t maxlen = 0; for (int x = 0; x < 8; ++x) maxlen = std::max(maxlen, xl[x].size());
  Tile nul; nul.cone = -1;
  std::vector<Tile> il(8 * maxlen, nul);
  for (int x = 0; x < 8; ++x) for (size_t s = 0; s < xl[x].size(); ++s) il[8 * s + x] = xl[x][s];
  printf("tiles %zu (per XCD:", dl.size()); for (int x = 0; x < 8; ++x) printf(" %zu", xl[x].size()); printf("), products %d, work %d clocks, persistent grid %d, cones %d, operand loads %s\n", nprod, work, grid, ncones, sc1 ? "sc1 (L1 bypass)" : "plain behind buffer_inv sc0 (WRONG form)");
  Tile *d_dl, *d_il; int *d_nt, *d_xoff; unsigned *d_sync, *d_wg; double* d_buf; unsigned long long* d_stale;
  hipMalloc(&d_dl, sizeof(Tile) * dl.size()); hipMemcpy(d_dl, dl.data(), sizeof(Tile) * dl.size(), hipMemcpyHostToDevice);
  hipMalloc(&d_il, sizeof(Tile) * il.size()); hipMemcpy(d_il, il.data(), sizeof(Tile) * il.size(), hipMemcpyHostToDevice);
  hipMalloc(&d_nt, sizeof(int) * ncones); hipMemcpy(d_nt, nt.data(), sizeof(int) * ncones, hipMemcpyHostToDevice);
  hipMalloc(&d_xoff, sizeof(int) * 9); hipMemcpy(d_xoff, xoff.data(), sizeof(int) * 9, hipMemcpyHostToDevice);
  hipMalloc(&d_sync, sizeof(unsigned) * (SYNC_DONE + ncones)); hipMalloc(&d_wg, sizeof(unsigned) * 8);
  hipMalloc(&d_buf, sizeof(double) * 2 * off); hipMalloc(&d_stale, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    float ms = 0.f;
    // persistent
    hipMemset(d_buf, 0, sizeof(double) * 2 * off); hipMemset(d_stale, 0, 8); hipMemset(d_wg, 0, 32);
    hipMemset(d_sync, 0, sizeof(unsigned) * (SYNC_DONE + ncones));
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    if (sc1) hipLaunchKernelGGL(k_dataflow<1>, dim3(grid), dim3(256), 0, 0, d_dl, d_nt, d_sync, d_buf, nprod, d_xoff, d_stale, d_wg);
    else hipLaunchKernelGGL(k_dataflow<0>, dim3(grid), dim3(256), 0, 0, d_dl, d_nt, d_sync, d_buf, nprod, d_xoff, d_stale, d_wg);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st = 0; unsigned wg[8], sy[160];
    hipMemcpy(&st, d_stale, 8, hipMemcpyDeviceToHost); hipMemcpy(wg, d_wg, 32, hipMemcpyDeviceToHost); hipMemcpy(sy, d_sync, sizeof(sy), hipMemcpyDeviceToHost);
    printf("persistent dependency-driven launch: %9.1f us total = %6.2f us per product-equivalent, stale wave-reads %llu, timeouts %u, workgroups per XCD", 1e3 * ms, 1e3 * ms / nprod, st, sy[129]);
    for (int x = 0; x < 8; ++x) printf(" %u", wg[x]);
    printf(", tickets"); for (int x = 0; x < 8; ++x) printf(" %u", sy[16 * x]);
    printf("\n");
    // launch per product
    hipMemset(d_buf, 0, sizeof(double) * 2 * off); hipMemset(d_stale, 0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int p = 0; p < nprod; ++p) hipLaunchKernelGGL(k_product, dim3((unsigned)il.size()), dim3(256), 0, 0, d_il, d_buf, p, d_stale);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&st, d_stale, 8, hipMemcpyDeviceToHost);
    printf("launch per product                 : %9.1f us total = %6.2f us per product, stale wave-reads %llu\n", 1e3 * ms, 1e3 * ms / nprod, st);
  }
  return 0;
}
