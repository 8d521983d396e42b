// batch.hip -- batches of INDEPENDENT small problems (BASELINE config 3: 1024 SOCPs with n = 500, m = 1000).
//
// The reference solves such a batch with one `optimize!` per model.  On the MI355X every problem gets ONE PERSISTENT
// WORKGROUP that runs the complete loop of src/solver.jl:137-176 for it -- projection, rho adaptation, CG solve, w update,
// termination check -- so that every synchronisation of the algorithm is a workgroup barrier instead of a kernel
// boundary and nothing is launch-bound; 1024 problems are 1024 resident workgroups (4 per CU).  All per-problem scalars
// (rho, CG residuals, counters, status) live with the problem: trajectories are those of 1024 separate solves, not of
// one block-diagonal solve.  The phases reuse the row lambdas / CSR-stream primitive of the large-problem path, the
// arithmetic per element is identical, reductions are single-workgroup fixed-order sums.
// Cone support in batch mode: ZeroSet, Nonnegatives, Box, SecondOrderCone and PsdCone / PsdConeTriangle of side <= 64 (round 4: side <= 16 by the
// wave-level Jacobi projection of psd16.h, 17 .. 64 by the workgroup-level block Jacobi of psdwg.h -- the routines the single-problem path uses for
// such cones, called from the problem's persistent workgroup: batches of small SDPs, src/convexset.jl:402-412 inside the composite projection
// :885-891) and ExponentialCone / PowerCone / their duals (cone3.h, one thread per cone); larger PSD cones take the single-problem path.  The
// infeasibility certificates run between persistent launches (k_batch_inf_capture / k_batch_inf_check below).
// Accelerator (round 4): with cosmo_hip_batch_set_accelerator the loop is the reference's accelerated loop (src/solver.jl:140-165,
// src/accelerator_interface.jl:58-116) per problem -- Anderson update / accelerate, safeguarding, deferred rho updates and certificates -- all inside
// the persistent workgroup (aa_pre / aa_declined / aa_reset below, shared by batch_admm_body and k_batch_admm_reg through an element visitor).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <chrono>
#include "device_utils.h"
#include "psd16.h"
#include "psdwg.h"
#include "cone3.h"

struct BCtl {                 // per problem, device resident
  int status; int n_rho_updates;
  long long iter, solves, kkt_iters_total;
  real rho, cost, r_prim, r_dual, max_norm_prim, max_norm_dual;
  real rho_updates[COSMO_HIP_MAX_RHO_UPDATES];
};

// Per-problem state of the Anderson accelerator (k_batch_admm*<..., AA = true>; csrc/anderson.hip is the single-problem form of the same
// algorithm).  R: the mem x mem triangular factor, column-major with leading dimension BAA_MEM.
#define BAA_MEM 16
struct BAa {
  int iter, init_phase, active, success;          // columns in memory; next update! only remembers (g, f); check_activation!; CA.was_successful
  int inf_due, need_inf, rho_due, fail_singular;  // ws.infeasibility_check_due; "run the certificates for this problem now" (host); ws.rho_update_due
  long long accelerated, accepted, declined, restarts, sg_iter;
  real nrm_f;
  real eta[BAA_MEM];
  real R[BAA_MEM * BAA_MEM];
};

struct BMat { const int* rowptr; const int* col; const real* val; const int* split; const int* rb;   // concatenated over problems
              const long long* nz_off; const int* rb_off; const int* nb; int nrows; int split_col; };

struct BatchDev {
  int nprob; int n; int m;
  BMat A, AT, PT;
  const real *q, *b, *Dinv, *Einv, *cinv;
  const uint32_t* meta; const real *box_l, *box_u; int nbox;
  int nsoc; const int *soc_off, *soc_dim;
  int npsd; const int *psd_off, *psd_d, *psd_kind;   // PSD cones of side 2..16: first row, side, COSMO_HIP_PSD_SQUARE / _TRIANGLE
  int psd_nws;                  // wave workspaces (psd16.h) available to a workgroup: waves 0 .. psd_nws-1 project the PSD cones
  // PSD cones of side 17 .. 64 (psdwg.h): one cone after the other by the whole workgroup; G = X + c I lives in global memory (ld x ncp per cone)
  int nmid; const int *mid_off, *mid_d, *mid_kind, *mid_ld, *mid_ncp; const long long* mid_goff;
  real* psdG; long long psdG_stride;      // per problem: sum over its mid cones of ld * ncp reals
  // ExponentialCone / PowerCone and their duals (cone3.h): first row, COSMO_HIP_EXP .. COSMO_HIP_DUAL_POW, alpha.  Handled by the kernel
  // instantiations that carry the PSD code (template flag PSD = "cones beyond Zero / Nonnegatives / Box / SecondOrderCone")
  int n3; const int *c3_off, *c3_kind; const real* c3_alpha;
  const int* rho_cls;
  real *w, *w_prev, *s, *mu, *s_tl, *ls_s, *y2, *tmp_m, *nu, *rho;
  real *ls_x, *x_tl, *rhs, *r, *u, *c;
  BCtl* ctl;
  real* inf_dy;                 // nprob * m: delta_y of the infeasibility certificates (captured between two launches)
  // Anderson accelerator (0 = none): G, Q are nprob slabs of (n + m) x aa_mem (column-major), f / f_last / g_last nprob x (n + m)
  int aa_mem; BAa* aa; real *aa_G, *aa_Q, *aa_f, *aa_fl, *aa_gl;
  const real* tol_table; long long tol_len;   // tol_constant / k^tol_exponent, k = 1.. (host libm, as the large path)
  // register kernel: COMPUTE assignment of the sparse passes inside the Krylov loop (see k_batch_admm_reg): thread t, slot j computes row
  // permA[k][j * 512 + t] of A (-1: none) and column permT[k][j * 512 + t] of [P | A'] -- rows / columns sorted by length, so that the
  // rows of a wave-step have similar lengths.  Null: every thread computes the rows it owns.
  const int *permA, *permT;
  const int *posN, *posM;       // nprob * n, nprob * m: position of an entry in the gathered LDS vectors (null: its index)
  const int* qposA;             // nprob * m: position of row i of A in the image's storage order (non-null iff the image is stored in sorted order)
  // sliced image of the register kernel (row_sliced): per problem, slot and thread the first entry of its row / column in the sliced arrays | length << 16;
  // pdiag: P is diagonal in every problem of the batch and lives in registers (nprob * n, 0 where a row of P is empty; pdiag_has: 1 where it has its entry)
  const uint32_t *slA, *slT; const real* pdiag; const unsigned char* pdiag_has; int sliced;
  int regcg;                    // LDS-image kernel with the extended cones (512 threads): Krylov vectors in registers (n <= 1024, m <= 2048; batch_admm_body)
};

struct BParams {
  real sigma, alpha, eps_abs, eps_rel, rho_min, rho_max, rho_eq, adapt_tol, obj_true, obj_true_tol;
  long long max_iter, max_adaptions;
  int check_termination, adaptive_rho, adaptive_rho_interval, unscale;
  // accelerated loop (AA kernels only)
  long long check_inf, aa_start_iter;     // check_infeasibility (0: no certificates in this launch); IterActivation
  int aa_min_mem, aa_safeguard;
  real aa_tau, aa_eta_max, aa_start_acc;  // safeguard_tol; eta_max; AccuracyActivation (< 0: unused)
};

__device__ __forceinline__ CsrView bview(const BMat& M, int k) {
  CsrView v;
  v.rowptr = M.rowptr + (long long)k * (M.nrows + 1);
  v.col = M.col + M.nz_off[k];
  v.val = M.val + M.nz_off[k];
  v.split = M.split ? M.split + (long long)k * M.nrows : nullptr;
  v.rb = M.rb + M.rb_off[k];
  v.nb = M.nb[k];
  v.nrows = M.nrows;
  v.split_col = M.split_col;
  return v;
}

__device__ __forceinline__ real proj_simple(real x, uint32_t meta, const real* bl, const real* bu) {
  const uint32_t kind = meta & 3u;
  if (kind == 0u) return x;
  if (kind == 1u) return 0.0;
  if (kind == 2u) return (x != x) ? x : ((x > R(0.0)) ? x : R(0.0));
  const uint32_t j = meta >> 2;
  const real l = bl[j], u = bu[j];
  return (x < l) ? l : ((x > u) ? u : x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Two ways of applying a problem's matrices inside its workgroup; both give every row to one thread that adds the row's
// products left to right (the order of Julia's CSC kernels), and both visit the rows in the order of the same tile
// descriptors, so per-thread accumulations and therefore all results are bit-identical between them.
//   StreamOps : values / indices stream from global memory tile by tile through a 16 KB LDS product buffer (any size).
//   LdsOps    : the whole problem is copied ONCE per launch into the workgroup's LDS -- A as (fp64 value, u16 column),
//               A' as (u16 position into A's values, u16 row) so the values are held once, P, the tile descriptors and
//               the two gathered vectors -- and every product of every iteration reads LDS only.  BASELINE config 3
//               (n = 500, m = 1000, nnz = 10 000) needs 157 KB of the CU's 160 KB.  HBM then only sees the iterates.
// ---------------------------------------------------------------------------------------------------------------------
template <int BS>
__device__ __forceinline__ real bsum(real v, real* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  real t = 0.0;
#pragma unroll
  for (int i = 0; i < BS / 64; ++i) t += red[i];
  return t;
}
// Block sum with ONE barrier: consecutive calls alternate between two sets of BS / 64 slots (ph flips).  The set written now was last read two
// calls ago, and every thread has passed the barrier of the call in between since then, so no barrier is needed in front of the writes.  Callers
// put a workgroup barrier between the last classic bsum / bmax on `red` and the first bsum_db (the classic ones start with their own barrier, so
// the other direction needs nothing).  Same additions in the same order as bsum.
template <int BS>
__device__ __forceinline__ real bsum_db(real v, real* red, int& ph) {
  v = wave_sum(v);
  real* r = red + ph * (BS / 64);
  ph ^= 1;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  real t = 0.0;
#pragma unroll
  for (int i = 0; i < BS / 64; ++i) t += r[i];
  return t;
}
template <int BS>
__device__ __forceinline__ real bmax(real v, real* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  real t = red[0];
#pragma unroll
  for (int i = 1; i < BS / 64; ++i) t = (red[i] > t || red[i] != red[i]) ? red[i] : t;
  return t;
}

#define PSD16_WS_STRIDE PSDWG_WS_STRIDE      // one per-wave workspace layout for both uses (psdwg.h: the psd16.h workspace + the panel's column list)
// the PSD cones (side <= 16) of a problem: waves 0 .. nws-1 take the cones round-robin, each on its own LDS workspace.  x = the projected slack
// (global memory, or its LDS staging copy in the register kernel).  Callers put a workgroup barrier in front and behind.
__device__ __forceinline__ void batch_project_psd(const BatchDev& D, real* x, unsigned char* ws_base, int wv, int lane) {
  if (wv >= D.psd_nws) return;
  const Psd16Ws ws = psd16_ws_at(ws_base + (size_t)wv * PSD16_WS_STRIDE);
  for (int cI = wv; cI < D.npsd; cI += D.psd_nws) {
    (void)psd16_wave(x + D.psd_off[cI], D.psd_d[cI], D.psd_kind[cI], ws, lane, 0, R(1.0), nullptr, nullptr);
    wave_lds_fence();                                   // the workspace is reused by this wave's next cone
  }
}

// ExponentialCone / PowerCone / duals of a problem: one thread per cone (src/convexset.jl:510-537, 626-684, 774-779; limits MAX_ITERS 100 / 20, tol 1e-8)
// (a real call, not inlined: the Newton / bisection loops with exp / log / pow would otherwise raise the register count of every kernel that carries
//  them -- the small-SDP instantiations are at the 256-VGPR limit already)
__device__ __attribute__((noinline)) void batch_project_cone3_one(real* p, int kind, real alpha) {
  cone3::V3 v{p[0], p[1], p[2]};
  (void)cone3::project_kind(v, kind, alpha, 100, 20, R(1e-8), R(1e-8));
  p[0] = v.x; p[1] = v.y; p[2] = v.z;
}
template <int BS>
__device__ __forceinline__ void batch_project_cone3(const BatchDev& D, real* x) {
  for (int c = threadIdx.x; c < D.n3; c += BS) batch_project_cone3_one(x + D.c3_off[c], D.c3_kind[c], D.c3_alpha[c]);
}

// the PSD cones of side 17 .. 64 of problem k: populate G = X + ||X||_F I, block-Jacobi sweeps (4 block pairs at most: 4 waves), column scaling +
// SYRK back into x.  Every thread of the workgroup takes part; ws_base needs 4 workspaces of PSDWG_WS_STRIDE bytes, red BS / 64 reals.
template <int BS>
__device__ __forceinline__ void batch_project_psd_mid(const BatchDev& D, int k, real* x, unsigned char* ws_base, real* red, int* any_rot) {
  for (int cI = 0; cI < D.nmid; ++cI) {
    real* g = D.psdG + (long long)k * D.psdG_stride + D.mid_goff[cI];
    const int d = D.mid_d[cI], kind = D.mid_kind[cI], ld = D.mid_ld[cI], ncp = D.mid_ncp[cI];
    real* xc = x + D.mid_off[cI];
    const real c = psdwg_populate<BS>(xc, d, kind, ld, ncp, R(1.0), 0, g, red);
    (void)psdwg_jacobi<4>(g, ld, ncp / 8, d, c, R(0.125), 0, ws_base, any_rot);
    __syncthreads();
    (void)psdwg_finish<BS>(xc, d, kind, ld, ncp, c, g, 0, red);
  }
}

struct StreamOps {
  CsrView A, AT, PT;
  real* lds; real* red;
  unsigned char* psd_ws;
  static constexpr bool in_lds = false;
  __device__ __forceinline__ real rowA(int, const real*) const { return R(0.0); }     // (register-CG form: LdsOps only)
  __device__ __forceinline__ real rowAT(int, const real*) const { return R(0.0); }
  __device__ __forceinline__ real rowP(int, const real*) const { return R(0.0); }
  __device__ __forceinline__ real* buf_n(real* g) const { return g; }       // vector the A / P products gather from
  __device__ __forceinline__ real* buf_m(real* g) const { return g; }       // vector the A' products gather from
  __device__ __forceinline__ const real* stage_n(const real* g) const { return g; }
  template <class F> __device__ __forceinline__ void rows_A(const real* x, F fn) {
    for (int t = 0; t < A.nb; ++t) csr_stream_tile(A, x, x, t, lds, red, fn);
  }
  template <class F> __device__ __forceinline__ void rows_AT(const real* y, F fn) {
    for (int t = 0; t < AT.nb; ++t) csr_stream_tile(AT, y, y, t, lds, red, fn);
  }
  template <class F> __device__ __forceinline__ void rows_PT(const real* x1, const real* x2, F fn) {
    for (int t = 0; t < PT.nb; ++t) csr_stream_tile(PT, x1, x2, t, lds, red, fn);
  }
};

// ---- hand-scheduled row loops of the register kernel's Krylov passes (round 5) --------------------------------------------------------------
// A row product needs per nonzero t:  P(t) an index load (u16 column position of A, or the packed u32 pair of A'), G(t) two b64 loads whose
// addresses come out of that index (value + gathered operand), M(t) multiply-add.  Written in C++ the compiler's scheduler moves the loads of
// trip t + 1 BEHIND the multiply of trip t to save registers, so every trip exposes an LDS round trip (index -> wait -> loads -> wait -> multiply: the
// ISA of both software-pipelined forms tried before).  Here the LDS reads and their waits are inline asm, which keeps program order:
//     trip k:   issue P(k+2)  |  s_waitcnt lgkmcnt(1): G(k) and P(k+1) have arrived (they were issued a full trip ago)  |  issue G(k+1)  |  M(k)
// ping-pong registers, two trips per loop body (a register rotation would have to wait for the loads in flight).  Same products, added in the
// same order by the same v_mul_f64 / v_add_f64 (the arithmetic stays C++): bit-identical to the plain loops.  Past the row's end the index
// pointer is clamped to the last nonzero: surplus loads read valid data that is never used.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)p; }     // low half of a flat address into LDS = its LDS byte offset
__device__ __forceinline__ void lds_read64(real& d, uint32_t addr) {
#if REAL_IS_FLOAT
  asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr));
#else
  asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr));
#endif
}
__device__ __forceinline__ void lds_read_u32(uint32_t& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_read_u16(uint32_t& d, uint32_t addr) { asm volatile("ds_read_u16 %0, %1" : "=v"(d) : "v"(addr)); }
// CONSTRAINT of the counted waits below: `s_waitcnt lgkmcnt(1)` means "the older LDS reads have arrived" only while NO scalar memory load is in
// flight -- s_load / s_buffer_load share the lgkm counter and return out of order, and the compiler's wait-count insertion does not see the counters of
// inline asm.  Two guards: (1) every hand-scheduled loop starts with lds_drain() (`s_waitcnt lgkmcnt(0)`: whatever the compiler issued in front of the loop --
// kernel-argument loads hoisted to that point, as found in the Float32 build of k_batch_admm_reg<512, 1, 2, true, false, false> -- has returned before the
// first asm read); (2) tests/test_isa_lgkm_waits.py disassembles both libraries at build time and fails if any counted lgkm wait can be reached with a
// scalar load outstanding, i.e. if a future compiler schedules one INTO a loop.
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)"); }
// wait until at most ONE LDS operation is outstanding; the operands tie the registers whose loads have arrived to this point of the program
__device__ __forceinline__ void lds_wait1(uint32_t& i, real& a, real& g) { asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(i), "+v"(a), "+v"(g)); }
__device__ __forceinline__ void lds_wait1(uint32_t& i) { asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(i)); }
__device__ __forceinline__ void lds_wait0(real& a, real& g) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(g)); }
#define RSH (REAL_IS_FLOAT ? 2 : 3)
// PAIR = true : A' row -- index array of packed u32 (value position | gather position << 16) at idx (byte address of the row's first entry)
// PAIR = false: A / P row -- index array of u16 gather positions at idx, the value of nonzero t sits at val + sizeof(real) * t (val = address of the row's first value)
template <bool PAIR>
__device__ __forceinline__ real row_pipe3(uint32_t idx, uint32_t val, const uint32_t gat, const int len) {
  if (len <= 0) return R(0.0);
  constexpr uint32_t ISZ = PAIR ? 4u : 2u;
  const uint32_t last = idx + ISZ * (uint32_t)(len - 1);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  auto load_idx = [&](uint32_t& d) { if (PAIR) lds_read_u32(d, idx); else lds_read_u16(d, idx); idx = (idx + ISZ < last) ? idx + ISZ : last; };
  auto load_ag = [&](real& a, real& g, uint32_t i, uint32_t vaddr) {
    if (PAIR) { lds_read64(a, val + ((i & 0xffffu) << RSH)); lds_read64(g, gat + ((i >> 16) << RSH)); }
    else { lds_read64(a, vaddr); lds_read64(g, gat + (i << RSH)); }
  };
  const uint32_t vlast = val + ((uint32_t)(len - 1) << RSH);
  uint32_t v1 = (len > 1) ? val + (1u << RSH) : vlast;           // value address of nonzero k + 1 (PAIR = false)
  lds_drain();
  load_idx(iA);                                                   // P(0)
  load_idx(iB);                                                   // P(1)
  lds_wait1(iA);
  load_ag(aA, gA, iA, val);                                       // G(0)
  int k = 0;
  for (;;) {
    load_idx(iA);                                                 // P(k+2)
    lds_wait1(iB, aA, gA);                                        // G(k), P(k+1) arrived
    load_ag(aB, gB, iB, v1);                                      // G(k+1)
    asm volatile("" : "+v"(aA), "+v"(gA));                       // (the product below is not scheduled in front of the loads above)
    v1 = (v1 + (1u << RSH) < vlast) ? v1 + (1u << RSH) : vlast;
    s += aA * gA;                                                 // M(k)
    if (++k >= len) break;
    load_idx(iB);                                                 // P(k+2)
    lds_wait1(iA, aB, gB);
    load_ag(aA, gA, iA, v1);
    asm volatile("" : "+v"(aB), "+v"(gB));
    v1 = (v1 + (1u << RSH) < vlast) ? v1 + (1u << RSH) : vlast;
    s += aB * gB;
    if (++k >= len) break;
  }
  lds_wait0(aA, gA);                                              // drain: nothing of this loop stays in flight behind it
  return s;
}

// ---- rows of many entries (round 6; the LONG instantiations of the register kernel) ---------------------------------------------------------------
// Every sparse pass gives a row to ONE thread, so a dense row of A -- the budget constraint sum(x) = 1 of a portfolio problem -- is walked serially in
// every Krylov iteration while 511 threads wait (profiles/r06_batch_dense_row_probe.txt: 3.9 -> 20 us per Krylov iteration at n = 300).  Here the lanes
// of the wave that holds such a row (>= LONG_ROW entries; the sorted assignment puts them in the first wave) walk it TOGETHER: lane l takes entries
// l, l + 64, ..., the partial sums are added by the wave butterfly, the owner keeps the result; the other rows of the wave then run on row_pipe3 as
// usual -- unless the wave holds so many long rows that walking them one after the other would cost more than its longest row.  Called by all lanes of a wave at the same point (no per-lane condition around the call: a lane without a row passes len = 0).  The sum of a
// long row is added in lane-strided instead of left-to-right order: a batch with such rows agrees with the other kernel forms to rounding, not bit for bit.
#define LONG_ROW 64
// which lanes of this wave hand their row to the whole wave (wave-uniform; computed ONCE per launch and slot: the rows of a slot do not change).  One thread
// per row costs the wave its LONGEST row (the lanes run side by side); together it costs the sum over the long rows of (entries / 64 + the butterfly,
// ~6 trips): a wave full of equally long rows -- a dense P-like block -- keeps one thread per row (mask 0).
__device__ __forceinline__ unsigned long long long_row_mask(const int len) {
  const bool is_long = len >= LONG_ROW;
  const unsigned long long todo = __ballot(is_long);
  if (!todo) return 0ull;
  int mx = is_long ? len : 0, sm = is_long ? (len + 63) / 64 + 6 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; sm += __shfl_xor(sm, o, 64); }
  return (sm < mx) ? todo : 0ull;
}
template <bool PAIR>
__device__ __forceinline__ real row_long_or_pipe3(uint32_t idx, uint32_t val, const uint32_t gat, const int len, unsigned long long todo) {
  constexpr uint32_t ISZ = PAIR ? 4u : 2u;
  const int lane = threadIdx.x & 63;
  const bool is_long = (todo >> lane) & 1ull;
  real mine = 0.0;
  while (todo) {                                                      // wave-uniform
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const uint32_t idx_s = (uint32_t)__shfl((int)idx, src, 64), val_s = (uint32_t)__shfl((int)val, src, 64);
    const int len_s = __shfl(len, src, 64);
    real part = 0.0;
    for (int t = lane; t < len_s; t += 64) {
      uint32_t e = 0; real a = 0.0, g = 0.0;
      lds_drain();
      if (PAIR) lds_read_u32(e, idx_s + ISZ * (uint32_t)t); else lds_read_u16(e, idx_s + ISZ * (uint32_t)t);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e));
      if (PAIR) { lds_read64(a, val_s + ((e & 0xffffu) << RSH)); lds_read64(g, gat + ((e >> 16) << RSH)); }
      else { lds_read64(a, val_s + ((uint32_t)t << RSH)); lds_read64(g, gat + (e << RSH)); }
      lds_wait0(a, g);
      part += a * g;
    }
    part = wave_sum(part);
    if (lane == src) mine = part;
  }
  const real s = row_pipe3<PAIR>(idx, val, gat, is_long ? 0 : len);
  return is_long ? mine : s;
}

// ---- sliced image of the register kernel (round 5, bench/lds_rowpipe_lab.hip) ------------------------------------------------------------------
// Rows sorted by length make the row-major stride of a wave EQUAL to the row length: with lengths 8, 12, 16 four to sixteen lanes of a half-wave read
// one bank pair, and the three address computations per nonzero were most of the instructions of a trip.  In the sliced image the 32 rows a half-wave
// works on in one trip are NEIGHBOURS: entry (trip, lane) of a slice of 32 threads sits at trip * 32 + lane (rows padded to the slice's longest: +5 % on
// config 3), so every value / index read of a trip is a linear wave access; the u16 entries of A / P hold the BYTE OFFSET of the gathered operand (the
// gathered vectors sit at fixed LDS addresses in front of the image: the loaded index is the address), the u32 entries of A' hold (LDS byte address of the
// value) | (position of the row << 21); pointers advance by immediate offsets; the trip count is the wave's longest row (scalar loop control) and a
// product past the lane's row end is not added.  Same products in the same order: bit-identical to row_pipe3.  Lab: A pass 3826 -> 2669, A' pass
// 5495 -> 4088 cycles (profiles/r05_lds_rowpipe_lab.txt).
#if !REAL_IS_FLOAT
#define SL_LANES 32
template <int OFF> __device__ __forceinline__ void lds_r64o(real& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_r32o(uint32_t& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_r16o(uint32_t& d, uint32_t addr) { asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
// A / P rows: ip = address of the lane's first u16 entry, vp = of its first value, GOFF = LDS address of the gathered vector, L = wave maximum of len
template <int GOFF>
__device__ __forceinline__ real row_sliced(uint32_t ip, uint32_t vp, const int len, const int L) {
  if (L <= 0) return R(0.0);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  lds_drain();
  lds_r16o<0>(iA, ip);
  lds_r16o<SL_LANES * 2>(iB, ip);
  lds_wait1(iA);
  lds_r64o<0>(aA, vp);
  lds_r64o<GOFF>(gA, iA);
  int k = 0;
  for (;;) {
    lds_r16o<SL_LANES * 4>(iA, ip);
    lds_wait1(iB, aA, gA);
    lds_r64o<SL_LANES * 8>(aB, vp);
    lds_r64o<GOFF>(gB, iB);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    lds_r16o<SL_LANES * 6>(iB, ip);
    lds_wait1(iA, aB, gB);
    lds_r64o<SL_LANES * 16>(aA, vp);
    lds_r64o<GOFF>(gA, iA);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += SL_LANES * 4u; vp += SL_LANES * 16u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA);
  asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
// A' rows: ip = address of the lane's first packed entry (value address | row position << 21)
template <int GOFF>
__device__ __forceinline__ real rowT_sliced(uint32_t ip, const int len, const int L) {
  if (L <= 0) return R(0.0);
  real s = 0.0;
  uint32_t iA = 0, iB = 0;
  real aA = 0.0, gA = 0.0, aB = 0.0, gB = 0.0;
  lds_drain();
  lds_r32o<0>(iA, ip);
  lds_r32o<SL_LANES * 4>(iB, ip);
  lds_wait1(iA);
  lds_r64o<0>(aA, iA & 0x3ffffu);
  lds_r64o<GOFF>(gA, iA >> 18);
  int k = 0;
  for (;;) {
    lds_r32o<SL_LANES * 8>(iA, ip);
    lds_wait1(iB, aA, gA);
    lds_r64o<0>(aB, iB & 0x3ffffu);
    lds_r64o<GOFF>(gB, iB >> 18);
    asm volatile("" : "+v"(aA), "+v"(gA));
    if (k < len) s += aA * gA;
    if (++k >= L) break;
    lds_r32o<SL_LANES * 12>(iB, ip);
    lds_wait1(iA, aB, gB);
    lds_r64o<0>(aA, iA & 0x3ffffu);
    lds_r64o<GOFF>(gA, iA >> 18);
    asm volatile("" : "+v"(aB), "+v"(gB));
    ip += SL_LANES * 8u;
    if (k < len) s += aB * gB;
    if (++k >= L) break;
  }
  lds_wait0(aA, gA);
  asm volatile("" : "+v"(aB), "+v"(gB));
  return s;
}
__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
#endif

// header of a problem's LDS image (built by build_lds_images): byte offsets from the start of the image
// oTpr: one u32 per nonzero of A' = (position into A's values) | (row, or its position in the gathered vector) << 16 -- ONE LDS read per nonzero instead
// of two u16 reads (round 5: the A' loop of the column pass issues 3 LDS instructions per nonzero instead of 4)
// oAord / oTord (round 6, 0 = none): the image is stored in the SORTED order of the register-CG compute assignment -- Arp is indexed by sorted position q and
// Aord[q] (u16) is the row stored there; Trp / Prp by the sorted position of the column, Tord[q] the column
struct LdsHdr { int nnzA, nnzP, nbA, nbAT, nbPT, oAval, oPval, oArp, oAcol, oTrp, oTpr, oPrp, oPcol, oRbA, oRbAT, oRbPT, bytes, oAord, oTord; };

template <int BS>
struct LdsOps {
  const real *Aval, *Pval;
  const unsigned short *Arp, *Acol, *Trp, *Prp, *Pcol;
  const uint32_t* Tpr;
  const int4 *rbA, *rbAT, *rbPT;
  const unsigned short *Aord, *Tord;          // stored-sorted image: row / column at a storage position (null: positions are indices)
  int nbA, nbAT, nbPT;
  real *xv, *tv, *red;
  unsigned char* psd_ws;
  int n;
  static constexpr bool in_lds = true;
  __device__ __forceinline__ int rowA_at(int q) const { return Aord ? (int)Aord[q] : q; }
  __device__ __forceinline__ int col_at(int q) const { return Tord ? (int)Tord[q] : q; }
  // whole rows, left to right, software-pipelined by one nonzero (the index / value loads of nonzero t + 1 go out with the gather of nonzero t): the
  // row functions of the register-CG form of the Krylov loop below (batch_admm_body, RCG).  (The hand-scheduled row_pipe3 was measured here too:
  // 262 vs 255 us per batch iteration, 443 vs 408 accelerated -- two short rows per thread do not amortise its prologue and drain; not used.)
  __device__ __forceinline__ real rowA(int r, const real* x) const {
    real s1 = 0.0;
    int t = Arp[r]; const int b2 = Arp[r + 1];
    if (t < b2) {
      real v = Aval[t]; int c = Acol[t];
      for (++t; t < b2; ++t) { const real vn = Aval[t]; const int cn = Acol[t]; s1 += v * x[c]; v = vn; c = cn; }
      s1 += v * x[c];
    }
    return s1;
  }
  __device__ __forceinline__ real rowAT(int r, const real* y) const {
    real s1 = 0.0;
    int t = Trp[r]; const int b2 = Trp[r + 1];
    if (t < b2) {
      uint32_t pr = Tpr[t];
      for (++t; t < b2; ++t) { const uint32_t prn = Tpr[t]; s1 += Aval[pr & 0xffffu] * y[pr >> 16]; pr = prn; }
      s1 += Aval[pr & 0xffffu] * y[pr >> 16];
    }
    return s1;
  }
  __device__ __forceinline__ real rowP(int r, const real* x) const {
    real s1 = 0.0;
    int t = Prp[r]; const int b2 = Prp[r + 1];
    if (t < b2) {
      real v = Pval[t]; int c = Pcol[t];
      for (++t; t < b2; ++t) { const real vn = Pval[t]; const int cn = Pcol[t]; s1 += v * x[c]; v = vn; c = cn; }
      s1 += v * x[c];
    }
    return s1;
  }
  __device__ __forceinline__ real rowA_b(int t, const int b2, const real* x) const {      // rows between two entry positions (bounds held by the caller)
    real s1 = 0.0;
    if (t < b2) {
      real v = Aval[t]; int c = Acol[t];
      for (++t; t < b2; ++t) { const real vn = Aval[t]; const int cn = Acol[t]; s1 += v * x[c]; v = vn; c = cn; }
      s1 += v * x[c];
    }
    return s1;
  }
  __device__ __forceinline__ real rowAT_b(int t, const int b2, const real* y) const {
    real s1 = 0.0;
    if (t < b2) {
      uint32_t pr = Tpr[t];
      for (++t; t < b2; ++t) { const uint32_t prn = Tpr[t]; s1 += Aval[pr & 0xffffu] * y[pr >> 16]; pr = prn; }
      s1 += Aval[pr & 0xffffu] * y[pr >> 16];
    }
    return s1;
  }
  __device__ __forceinline__ real rowP_b(int t, const int b2, const real* x) const {
    real s1 = 0.0;
    if (t < b2) {
      real v = Pval[t]; int c = Pcol[t];
      for (++t; t < b2; ++t) { const real vn = Pval[t]; const int cn = Pcol[t]; s1 += v * x[c]; v = vn; c = cn; }
      s1 += v * x[c];
    }
    return s1;
  }
  __device__ __forceinline__ real* buf_n(real*) const { return xv; }
  __device__ __forceinline__ real* buf_m(real*) const { return tv; }
  __device__ __forceinline__ const real* stage_n(const real* g) const {     // copy a global n-vector into the LDS gather buffer
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += BS) xv[i] = g[i];
    __syncthreads();
    return xv;
  }
  template <class F> __device__ __forceinline__ void rows_A(const real* x, F fn) {
    __syncthreads();
    for (int t = 0; t < nbA; ++t) {
      const int4 d = rbA[t];
      if (d.w - d.z <= COSMO_NNZ_PER_BLOCK) {
        for (int r = d.x + threadIdx.x; r < d.y; r += BS) {
          real s1 = 0.0;
          const int a = Arp[r], b = Arp[r + 1];
          for (int k = a; k < b; ++k) s1 += Aval[k] * x[Acol[k]];
          fn(rowA_at(r), s1, 0.0);
        }
      } else {                                                               // a single long row: strided partials, block sum
        const int r = d.x;
        real s1 = 0.0, s2 = 0.0;
        for (int k = d.z + threadIdx.x; k < d.w; k += BS) s1 += Aval[k] * x[Acol[k]];
        s1 = bsum<BS>(s1, red); s2 = bsum<BS>(s2, red);
        if (threadIdx.x == 0) fn(rowA_at(r), s1, s2);
      }
    }
    __syncthreads();
  }
  template <class F> __device__ __forceinline__ void rows_AT(const real* y, F fn) {
    __syncthreads();
    for (int t = 0; t < nbAT; ++t) {
      const int4 d = rbAT[t];
      if (d.w - d.z <= COSMO_NNZ_PER_BLOCK) {
        for (int r = d.x + threadIdx.x; r < d.y; r += BS) {
          real s1 = 0.0;
          const int a = Trp[r], b = Trp[r + 1];
          for (int k = a; k < b; ++k) { const uint32_t pr = Tpr[k]; s1 += Aval[pr & 0xffffu] * y[pr >> 16]; }
          fn(col_at(r), s1, 0.0);
        }
      } else {
        const int r = d.x;
        real s1 = 0.0, s2 = 0.0;
        for (int k = d.z + threadIdx.x; k < d.w; k += BS) { const uint32_t pr = Tpr[k]; s1 += Aval[pr & 0xffffu] * y[pr >> 16]; }
        s1 = bsum<BS>(s1, red); s2 = bsum<BS>(s2, red);
        if (threadIdx.x == 0) fn(col_at(r), s1, s2);
      }
    }
    __syncthreads();
  }
  template <class F> __device__ __forceinline__ void rows_PT(const real* x1, const real* x2, F fn) {
    __syncthreads();
    for (int t = 0; t < nbPT; ++t) {
      const int4 d = rbPT[t];
      if (d.w - d.z <= COSMO_NNZ_PER_BLOCK) {
        for (int r = d.x + threadIdx.x; r < d.y; r += BS) {
          real s1 = 0.0, s2 = 0.0;
          const int pa = Prp[r], pb = Prp[r + 1];
          for (int k = pa; k < pb; ++k) s1 += Pval[k] * x1[Pcol[k]];
          const int a = Trp[r], b = Trp[r + 1];
          for (int k = a; k < b; ++k) { const uint32_t pr = Tpr[k]; s2 += Aval[pr & 0xffffu] * x2[pr >> 16]; }
          fn(col_at(r), s1, s2);
        }
      } else {
        const int r = d.x;
        const int pa = Prp[r], lp = Prp[r + 1] - pa, a = Trp[r], lt = Trp[r + 1] - a;
        real s1 = 0.0, s2 = 0.0;
        for (int k = threadIdx.x; k < lp + lt; k += BS) {
          if (k < lp) s1 += Pval[pa + k] * x1[Pcol[pa + k]];
          else { const uint32_t pr = Tpr[a + (k - lp)]; s2 += Aval[pr & 0xffffu] * x2[pr >> 16]; }
        }
        s1 = bsum<BS>(s1, red); s2 = bsum<BS>(s2, red);
        if (threadIdx.x == 0) fn(col_at(r), s1, s2);
      }
    }
    __syncthreads();
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// The accelerator inside a problem's workgroup (shared by batch_admm_body and the register kernel).  `each(fn)` visits the elements of
// w = [x ; rows] this THREAD owns as fn(e, w_e, w_prev_e) (references: global memory in batch_admm_body, registers in k_batch_admm_reg); the
// accelerator's own vectors (G, Q, f, f_last, g_last) live in global memory, element e of each touched only by its owner, so no barrier is needed
// between passes beyond those of the block sums.  All scalars (AaRegs) are block-uniform.
// ---------------------------------------------------------------------------------------------------------------------
struct AaRegs {
  int iter = 0, init = 1, active = 0, success = 0, inf_due = 0, rho_due = 0, need_inf = 0;
  long long n_acc = 0, n_ok = 0, n_decl = 0, n_rst = 0, sg = 0;
  real nrmf = 0.0;
};
struct AaMem { BAa* aa; real *G, *Q, *f, *fl, *gl; int N, mem; };

__device__ __forceinline__ AaMem aa_mem_of(const BatchDev& D, int k) {
  AaMem M;
  M.N = D.n + D.m; M.mem = D.aa_mem; M.aa = D.aa + k;
  M.G = D.aa_G + (long long)k * M.N * M.mem; M.Q = D.aa_Q + (long long)k * M.N * M.mem;
  const long long onm = (long long)k * M.N;
  M.f = D.aa_f + onm; M.fl = D.aa_fl + onm; M.gl = D.aa_gl + onm;
  return M;
}
__device__ __forceinline__ void aa_load(AaRegs& S, const BAa* a) {
  S.iter = a->iter; S.init = a->init_phase; S.active = a->active; S.success = a->success; S.inf_due = a->inf_due; S.rho_due = a->rho_due; S.need_inf = 0;
  S.n_acc = a->accelerated; S.n_ok = a->accepted; S.n_decl = a->declined; S.n_rst = a->restarts; S.sg = a->sg_iter; S.nrmf = a->nrm_f;
}
__device__ __forceinline__ void aa_store(const AaRegs& S, BAa* a) {
  a->iter = S.iter; a->init_phase = S.init; a->active = S.active; a->success = S.success; a->inf_due = S.inf_due; a->rho_due = S.rho_due; a->need_inf = S.need_inf;
  a->accelerated = S.n_acc; a->accepted = S.n_ok; a->declined = S.n_decl; a->restarts = S.n_rst; a->sg_iter = S.sg; a->nrm_f = S.nrmf;
}

// acceleration_pre! (accelerator_interface.jl:58-76): check_activation!, CA.update!(w, w_prev), CA.accelerate!(w).  update!: f = x - g;
// G_j = g - g_last; v = f - f_last, modified Gram-Schmidt of v against Q_0..Q_{j-1} -> R[0..j, j], Q_j.  accelerate!: eta = R \ (Q' f);
// w -= G eta unless R is singular / not finite or ||eta||_2 > eta_max.  (g = w, x = w_prev.)
template <int BS, class Each>
__device__ __forceinline__ void aa_pre(AaRegs& S, const AaMem& M, const BParams& P, long long it, Each each, real* red) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = M.N;
  S.success = 0;
  if (!S.active && !(P.aa_start_acc >= R(0.0)) && it >= P.aa_start_iter) S.active = 1;
  if (!S.active) return;
  if (S.init) {
    each([&](int e, real& we, real& wpe) { const real gi = we; const real fi = wpe - gi; M.f[e] = fi; M.gl[e] = gi; M.fl[e] = fi; });
    S.init = 0;
  } else {
    int j = S.iter % M.mem;
    if (j == 0 && S.iter != 0) { S.iter = 0; S.n_rst += 1; }             // RestartedMemory: every column is rewritten before it is read again
    real* Gj = M.G + (long long)j * N; real* v = M.Q + (long long)j * N;
    real acc = 0.0;
    each([&](int e, real& we, real& wpe) {
      const real gi = we; const real fi = wpe - gi;
      M.f[e] = fi; Gj[e] = gi - M.gl[e];
      const real vi = fi - M.fl[e];
      v[e] = vi; M.gl[e] = gi; M.fl[e] = fi;
      acc += (j == 0) ? vi * vi : M.Q[e] * vi;
    });
    real rv = bsum<BS>(acc, red);
    for (int i = 0; i < j; ++i) {
      const bool last = (i + 1 == j);
      const real* Qi = M.Q + (long long)i * N; const real* Qn = M.Q + (long long)(last ? i : i + 1) * N;
      const real r = rv;
      if (tid == 0) M.aa->R[j * BAA_MEM + i] = r;
      acc = 0.0;
      each([&](int e, real&, real&) { const real vi = v[e] - r * Qi[e]; v[e] = vi; acc += last ? vi * vi : Qn[e] * vi; });
      rv = bsum<BS>(acc, red);
    }
    const real nv = sqrt(rv);
    if (tid == 0) M.aa->R[j * BAA_MEM + j] = nv;
    each([&](int e, real&, real&) { v[e] = v[e] / nv; });
    S.iter += 1;
  }
  const int l = S.iter < M.mem ? S.iter : M.mem;
  if (l < P.aa_min_mem) return;
  real myrhs = 0.0;                                                       // lane c (of every wave) keeps (Q' f)[c]
  for (int c = 0; c < l; ++c) {
    const real* Qc = M.Q + (long long)c * N;
    real acc = 0.0;
    each([&](int e, real&, real&) { acc += Qc[e] * M.f[e]; });
    const real r = bsum<BS>(acc, red);
    if (lane == c) myrhs = r;
  }
  { real acc = 0.0;
    each([&](int e, real&, real&) { const real fi = M.f[e]; acc += fi * fi; });
    S.nrmf = sqrt(bsum<BS>(acc, red)); }
  __syncthreads();                                                        // R[., j] of thread 0 is visible to wave 0
  if (wv == 0) {                                                          // back substitution on one wave: lane i owns row i of R
    const bool mine = lane < l;
    real row[BAA_MEM];
    real diag = 1.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < BAA_MEM; ++c) {
      const bool use = mine && c >= lane && c < l;
      row[c] = use ? M.aa->R[c * BAA_MEM + lane] : R(0.0);
      if (use && !(fabs(row[c]) <= REAL_MAX)) bad = true;
      if (use && c == lane) diag = row[c];
    }
    if (mine && diag == R(0.0)) bad = true;
    int ok = __any(bad ? 1 : 0) ? 0 : 1;
    const int singular = ok ? 0 : 1;
    if (ok) {
      real sacc = myrhs, nrm2 = 0.0;
#pragma unroll
      for (int c = BAA_MEM - 1; c >= 0; --c) {
        if (c < l) {
          const real e_c = __shfl(sacc / diag, c, 64);
          nrm2 += e_c * e_c;
          if (lane < c) sacc -= row[c] * e_c;
          if (lane == 0) M.aa->eta[c] = e_c;
        }
      }
      if (!(sqrt(nrm2) <= P.aa_eta_max)) ok = 0;                          // also catches NaN
    }
    if (lane == 0) { M.aa->success = ok; if (singular) M.aa->fail_singular += 1; }
  }
  __syncthreads();
  S.success = M.aa->success;
  if (S.success) {
    real eta[BAA_MEM];
#pragma unroll
    for (int c = 0; c < BAA_MEM; ++c) eta[c] = (c < l) ? M.aa->eta[c] : R(0.0);
    each([&](int e, real& we, real&) {
      real sacc = 0.0;
#pragma unroll
      for (int c = 0; c < BAA_MEM; ++c) if (c < l) sacc += M.G[(long long)c * N + e] * eta[c];
      we = we - sacc;
    });
  }
  __syncthreads();
}
// acceleration_post! part 1 (accelerator_interface.jl:85-100, 123-126): f = w_prev - w of the accelerated point; true = declined (the caller resets
// and re-does the ADMM step)
template <int BS, class Each>
__device__ __forceinline__ bool aa_declined(AaRegs& S, const AaMem& M, const BParams& P, Each each, real* red) {
  real acc = 0.0;
  each([&](int e, real& we, real& wpe) { const real d = wpe - we; M.f[e] = d; acc += d * d; });
  const real nrm_acc = sqrt(bsum<BS>(acc, red));
  return nrm_acc > S.nrmf * P.aa_tau;
}
// reset_accelerated_vector! (accelerator_interface.jl:129-134)
template <class Each>
__device__ __forceinline__ void aa_reset(const AaMem& M, Each each) {
  each([&](int e, real& we, real& wpe) { const real g = M.gl[e]; we = g; wpe = g; });
}

// One workgroup = one problem.  Runs iterations until a status is decided or `iter_target` iterations are done.
// PSD: the batch has PsdCone / PsdConeTriangle cones of side 2..16.  A template parameter, not a run-time test: the wave-level Jacobi of psd16.h
// inlined into these kernels costs 55-60 VGPRs (the register kernel <512, 1, 2> went from 229 to 256 + spills), which batches without such
// cones -- BASELINE config 3 -- must not pay.
// AA: the accelerated loop (src/solver.jl:140-165 with an AndersonAccelerator; csrc/anderson.hip + optimize_accelerated of api.hip are the
// single-problem form).  Every inner product of the accelerator is a block sum of the workgroup, every decision (success of the least-squares
// step, safeguarding, deferred rho update / infeasibility check) is taken by the workgroup for its problem.
#ifndef COSMO_LDSCG_HANDPIPE
#define COSMO_LDSCG_HANDPIPE 1          // lab builds: 0 = the compiled one-stage pipelined row loops in the register-CG form of the LDS-image kernel
#endif
template <int BS, bool PSD, bool AA, class Ops>
__device__ __forceinline__ void batch_admm_body(const BatchDev& D, const BParams& P, long long iter_target, int do_init, Ops& ops, real* red, const int k) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = D.n, m = D.m;
  BCtl* ctl = D.ctl + k;
  const long long on = (long long)k * n, om = (long long)k * m, onm = (long long)k * (n + m);
  const real *q = D.q + on, *b = D.b + om, *Dinv = D.Dinv + on, *Einv = D.Einv + om;
  const real cinv = D.cinv[k];
  const real *bl = D.box_l + (long long)k * D.nbox, *bu = D.box_u + (long long)k * D.nbox;
  const int* cls = D.rho_cls + om;
  real *w = D.w + onm, *w_prev = D.w_prev + onm, *s = D.s + om, *mu = D.mu + om, *s_tl = D.s_tl + om;
  real *ls_s = D.ls_s + om, *nu = D.nu + om, *rho = D.rho + om;
  real *ls_x = D.ls_x + on, *x_tl = D.x_tl + on, *rhs = D.rhs + on, *r = D.r + on, *c = D.c + on;
  // vectors the sparse products GATHER from: global arrays for StreamOps, the two LDS buffers for LdsOps
  real* const u = ops.buf_n(D.u + on);                 // CG direction
  real* const y2 = ops.buf_m(D.y2 + om);               // rho .* ls_s (dead once the rhs is formed)
  real* const tmp_m = ops.buf_m(D.tmp_m + om);         // rho .* (A v)
  real* const mu_g = ops.buf_m(mu);                    // mu for the dual residual

  // ---- admm_x! + admm_w! (solver.jl:32-65) with the CG reduced solve (kktsolver_indirect.jl:36-88) -------------------
  // Register-CG form of the whole reduced solve (LDS-image kernel of the batches with PSD / exponential / power cones and of their accelerated runs;
  // round 5: the Krylov loop, round 6: every sparse pass of the solve).  Thread t COMPUTES the rows ra[] of A and the column ct[] of [P | A'] of the
  // length-sorted assignment (D.permA / D.permT in the <2, 4>-slot layout of this kernel, build_lds_images; index order without the tables:
  // COSMO_HIP_BATCH_LDSCG_SORTED=0) -- the rows of a wave-step have similar lengths -- and HOLDS elements ct[] of rhs, r, u, x_tl, c in registers for
  // the solve: the ownership is local to the solve, so c never travels and global memory sees x_tl once at the start and once at the end.  The bounds
  // of those rows / columns are read once per solve; the A and A' rows run on the hand-scheduled two-stage pipeline row_pipe3; the block sums take one
  // barrier each (bsum_db).  Round 6 also moved the three passes in front of the loop (rhs = A' y2 + ls_x, tmp = rho .* A x_tl, r = rhs - M x_tl)
  // and the pass behind it (A x_tl for nu / s_tl / w) from the tile loops of LdsOps (~10 us each on config 3) onto the same rows: ~1.5 us each.  Same
  // left-to-right row sums everywhere; the block sums add the per-element terms in the order of this assignment (trajectories agree with the
  // streaming kernel to 1e-9 in tight-CG mode; COSMO_HIP_BATCH_LDSCG=0 restores the generic form below).
  constexpr bool RCG = Ops::in_lds && PSD && BS == 512;
  auto solve_and_update_rcg = [&]() {
    if constexpr (RCG) {
    constexpr int RJN = 2, RJM = 4;
    int ra[RJM], ct[RJN];
    uint32_t kA[RJM], kT[RJN], kP[RJN];                                  // first entry | entries << 16 of the row of A, the columns of A' and P (u16 each: the image's limits)
    real rR[RJN], uR[RJN], xR[RJN], cR[RJN], rhoR[RJM];
    for (int i = tid; i < n + m; i += BS) {                             // rhs of the KKT system + y2 = rho .* ls_s  (index owners)
      if (i < n) ls_x[i] = P.sigma * w[i] - q[i];
      else { const int rr = i - n; const real v = (b[rr] - R(2.0) * s[rr]) + w[i]; ls_s[rr] = v; y2[rr] = rho[rr] * v; }
    }
#pragma unroll
    for (int j = 0; j < RJM; ++j) {
      const int i = tid + BS * j;
      const int rr_ = D.permA ? D.permA[((long long)k * RJM + j) * BS + tid] : (i < m ? i : -1);
      ra[j] = rr_;
      rhoR[j] = rr_ >= 0 ? rho[rr_] : R(1.0);
      const int qa = ops.Aord ? BS * j + ((j & 1) ? BS - 1 - tid : tid) : rr_;       // stored-sorted image: Arp is indexed by the position of the compute assignment
      const int a0 = rr_ >= 0 ? (int)ops.Arp[qa] : 0, a1 = rr_ >= 0 ? (int)ops.Arp[qa + 1] : 0;
      kA[j] = (uint32_t)a0 | ((uint32_t)(a1 - a0) << 16);
    }
#pragma unroll
    for (int j = 0; j < RJN; ++j) {
      const int i = tid + BS * j;
      const int cc_ = D.permT ? D.permT[((long long)k * RJN + j) * BS + tid] : (i < n ? i : -1);
      ct[j] = cc_;
      const bool ok = cc_ >= 0;
      xR[j] = ok ? x_tl[cc_] : R(0.0); uR[j] = 0.0; cR[j] = 0.0; rR[j] = 0.0;
      if (ok) u[cc_] = xR[j];                                            // the n-vector buffer gathers x_tl first
      const int qc = ops.Tord ? BS * j + tid : cc_;
      const int t0 = ok ? (int)ops.Trp[qc] : 0, t1 = ok ? (int)ops.Trp[qc + 1] : 0;
      kT[j] = (uint32_t)t0 | ((uint32_t)(t1 - t0) << 16);
      const int p0 = ok ? (int)ops.Prp[qc] : 0, p1 = ok ? (int)ops.Prp[qc + 1] : 0;
      kP[j] = (uint32_t)p0 | ((uint32_t)(p1 - p0) << 16);
    }
    const uint32_t lA_val = lds_addr_of(ops.Aval), lA_col = lds_addr_of(ops.Acol), lT_pr = lds_addr_of(ops.Tpr), l_xv = lds_addr_of(u), l_tv = lds_addr_of(tmp_m);
    (void)lA_val; (void)lA_col; (void)lT_pr; (void)l_xv; (void)l_tv;
    auto rowA_c = [&](int j) -> real {                                   // row ra[j] of A times the n-vector buffer
      return COSMO_LDSCG_HANDPIPE ? row_pipe3<false>(lA_col + 2u * (kA[j] & 0xffffu), lA_val + ((kA[j] & 0xffffu) << RSH), l_xv, (int)(kA[j] >> 16))
                                  : ops.rowA_b((int)(kA[j] & 0xffffu), (int)((kA[j] & 0xffffu) + (kA[j] >> 16)), u);
    };
    auto colT_c = [&](int j) -> real {                                   // column ct[j] of A (row of A') times the m-vector buffer
      const int t0 = (int)(kT[j] & 0xffffu);
      return COSMO_LDSCG_HANDPIPE ? row_pipe3<true>(lT_pr + 4u * (uint32_t)t0, lA_val, l_tv, (int)(kT[j] >> 16)) : ops.rowAT_b(t0, t0 + (int)(kT[j] >> 16), tmp_m);
    };
    auto colP_c = [&](int j) -> real { const int p0 = (int)(kP[j] & 0xffffu); return ops.rowP_b(p0, p0 + (int)(kP[j] >> 16), u); };
    int ph = 0;
    __syncthreads();                                                     // y2, the gathered x_tl and ls_x are visible; the classic block sums on `red` are behind us (bsum_db)
    real acc = 0.0;
    real rhsR[RJN];
#pragma unroll
    for (int j = 0; j < RJN; ++j) {
      rhsR[j] = 0.0;
      if (ct[j] >= 0) { const real v = (colT_c(j) + R(0.0)) + ls_x[ct[j]]; rhsR[j] = v; acc += v * v; }
    }
    const real bb = bsum_db<BS>(acc, red, ph);                           // (its barrier also orders the reads of y2 before the writes of tmp_m below: one buffer)
    real tmpv[RJM];
#pragma unroll
    for (int j = 0; j < RJM; ++j) tmpv[j] = (ra[j] >= 0) ? (rowA_c(j) + R(0.0)) * rhoR[j] : R(0.0);
#pragma unroll
    for (int j = 0; j < RJM; ++j) if (ra[j] >= 0) tmp_m[ra[j]] = tmpv[j];
    __syncthreads();
    acc = 0.0;
#pragma unroll
    for (int j = 0; j < RJN; ++j) {
      if (ct[j] >= 0) { const real cj = colP_c(j) + (P.sigma * xR[j] + (colT_c(j) + R(0.0))); const real rj = rhsR[j] - cj; rR[j] = rj; acc += rj * rj; }
    }
    real rr = bsum_db<BS>(acc, red, ph);
    const long long ks = ctl->solves;                                    // iteration_counter - 1
    const real tol_k = D.tol_table[ks < D.tol_len ? ks : D.tol_len - 1];
    const real tol = tol_k / sqrt(bb);
    real res = sqrt(rr), prev = 1.0;
    int kk = 0;
    while (kk < n && !(res <= tol)) {                                    // cg! (IterativeSolvers v0.9), maxiter = n
      const real beta = (res * res) / (prev * prev);
#pragma unroll
      for (int j = 0; j < RJN; ++j) { uR[j] = rR[j] + beta * ((kk == 0) ? R(0.0) : uR[j]); if (ct[j] >= 0) u[ct[j]] = uR[j]; }   // (u was last read before the block sums above)
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RJM; ++j) tmpv[j] = (ra[j] >= 0) ? rowA_c(j) * rhoR[j] : R(0.0);
#pragma unroll
      for (int j = 0; j < RJM; ++j) if (ra[j] >= 0) tmp_m[ra[j]] = tmpv[j];     // (tmp_m was last read before the block sums of the previous iteration)
      __syncthreads();
      acc = 0.0;
#pragma unroll
      for (int j = 0; j < RJN; ++j) {
        if (ct[j] >= 0) { const real vj = uR[j]; const real cj = colP_c(j) + (P.sigma * vj + colT_c(j)); cR[j] = cj; acc += vj * cj; }
      }
      const real uc = bsum_db<BS>(acc, red, ph);
      const real a = (res * res) / uc;
      acc = 0.0;
#pragma unroll
      for (int j = 0; j < RJN; ++j) { xR[j] = xR[j] + a * uR[j]; const real ri = rR[j] - a * cR[j]; rR[j] = ri; acc += ri * ri; }
      rr = bsum_db<BS>(acc, red, ph);
      prev = res; res = sqrt(rr); ++kk;
    }
    // nu = rho (A x_tl - ls_s) ; s_tl ; w update: A x_tl by the computing threads, handed to the index owners through the m-vector buffer
#pragma unroll
    for (int j = 0; j < RJN; ++j) if (ct[j] >= 0) { x_tl[ct[j]] = xR[j]; u[ct[j]] = xR[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RJM; ++j) tmpv[j] = (ra[j] >= 0) ? (rowA_c(j) + R(0.0)) : R(0.0);
#pragma unroll
    for (int j = 0; j < RJM; ++j) if (ra[j] >= 0) tmp_m[ra[j]] = tmpv[j];
    __syncthreads();
    for (int i = tid; i < n + m; i += BS) {
      if (i < n) { const real wv2 = w[i]; w[i] = wv2 + P.alpha * (u[i] - wv2); }
      else {
        const int row = i - n;
        const real rh = rho[row]; const real nv = (tmp_m[row] - ls_s[row]) * rh; nu[row] = nv;
        const real sv = s[row], wv2 = w[i]; const real st = (R(2.0) * sv - wv2) - nv / rh; s_tl[row] = st;
        w[i] = wv2 + P.alpha * (st - sv);
      }
    }
    __syncthreads();
    if (tid == 0) { ctl->solves = ks + 1; ctl->kkt_iters_total += kk; }
    __syncthreads();
    }
  };
  auto solve_and_update = [&]() {
    if constexpr (RCG) { if (D.regcg != 0) { solve_and_update_rcg(); return; } }
    for (int i = tid; i < n + m; i += BS) {                             // rhs of the KKT system + y2 = rho .* ls_s
      if (i < n) ls_x[i] = P.sigma * w[i] - q[i];
      else { const int rr = i - n; const real v = (b[rr] - R(2.0) * s[rr]) + w[i]; ls_s[rr] = v; y2[rr] = rho[rr] * v; }
    }
    __syncthreads();
    real acc = 0.0;
    ops.rows_AT(y2, [&](int row, real s1, real s2) {
      const real v = (s1 + s2) + ls_x[row]; rhs[row] = v; acc += v * v; });
    const real bb = bsum<BS>(acc, red);
    const real* xs = ops.stage_n(x_tl);
    ops.rows_A(xs, [&](int row, real s1, real s2) { tmp_m[row] = (s1 + s2) * rho[row]; });
    __syncthreads();
    acc = 0.0;
    ops.rows_PT(xs, tmp_m, [&](int row, real s1, real s2) {
      const real cj = s1 + (P.sigma * xs[row] + s2); const real rj = rhs[row] - cj; r[row] = rj; acc += rj * rj; });
    real rr = bsum<BS>(acc, red);
    const long long ks = ctl->solves;                                    // iteration_counter - 1
    const real tol_k = D.tol_table[ks < D.tol_len ? ks : D.tol_len - 1];
    const real tol = tol_k / sqrt(bb);
    real res = sqrt(rr), prev = 1.0;
    int kk = 0;
    while (kk < n && !(res <= tol)) {                                    // cg! (IterativeSolvers v0.9), maxiter = n
      const real beta = (res * res) / (prev * prev);
      for (int i = tid; i < n; i += BS) u[i] = r[i] + beta * ((kk == 0) ? R(0.0) : u[i]);
      __syncthreads();
      ops.rows_A(u, [&](int row, real s1, real s2) { tmp_m[row] = (s1 + s2) * rho[row]; });
      __syncthreads();
      acc = 0.0;
      ops.rows_PT(u, tmp_m, [&](int row, real s1, real s2) {
        const real vj = u[row]; const real cj = s1 + (P.sigma * vj + s2); c[row] = cj; acc += vj * cj; });
      const real uc = bsum<BS>(acc, red);
      const real a = (res * res) / uc;
      acc = 0.0;
      for (int i = tid; i < n; i += BS) {
        x_tl[i] = x_tl[i] + a * u[i];
        const real ri = r[i] - a * c[i]; r[i] = ri; acc += ri * ri;
      }
      rr = bsum<BS>(acc, red);
      prev = res; res = sqrt(rr); ++kk;
    }
    __syncthreads();
    // nu = rho (A x_tl - ls_s) ; s_tl ; w update
    const real* xe = ops.stage_n(x_tl);
    ops.rows_A(xe, [&](int row, real s1, real s2) {
      const real rh = rho[row]; const real nv = ((s1 + s2) - ls_s[row]) * rh; nu[row] = nv;
      const real sv = s[row], wv2 = w[n + row]; const real st = (R(2.0) * sv - wv2) - nv / rh; s_tl[row] = st;
      w[n + row] = wv2 + P.alpha * (st - sv); });
    for (int i = tid; i < n; i += BS) { const real wv2 = w[i]; w[i] = wv2 + P.alpha * (x_tl[i] - wv2); }
    __syncthreads();
    if (tid == 0) { ctl->solves = ks + 1; ctl->kkt_iters_total += kk; }
    __syncthreads();
  };

  // ---- residuals (residuals.jl:30-96,143-147); x = w_prev[1:n], mu recovered on the fly ------------------------------
  real rp, mp, rd, md, cost;
  auto residuals = [&](bool unscale) {
    real a_rp = 0.0, a_mp = 0.0;
    const real* xp = ops.stage_n(w_prev);
    ops.rows_A(xp, [&](int row, real s1, real s2) {
      const real ax = s1 + s2, sv = s[row], bv = b[row];
      const real muv = rho[row] * (w_prev[n + row] - sv);
      mu[row] = muv;
      if (Ops::in_lds) mu_g[row] = muv;
      real rv = ax + sv; rv = rv - bv;
      const real e = unscale ? Einv[row] : 1.0;
      if (unscale) rv = rv * e;
      a_rp = amax(a_rp, rv);
      a_mp = amax(a_mp, unscale ? ax * e : ax); a_mp = amax(a_mp, unscale ? sv * e : sv); a_mp = amax(a_mp, unscale ? bv * e : bv); });
    rp = bmax<BS>(a_rp, red); mp = bmax<BS>(a_mp, red);
    __syncthreads();
    real a_rd = 0.0, a_md = 0.0, xpx = 0.0, qx = 0.0;
    ops.rows_PT(xp, mu_g, [&](int row, real px, real atm) {
      const real xv = xp[row], qv = q[row];
      real rv = px + qv; rv = rv - atm;
      real a = px, bq = qv, cm = atm;
      if (unscale) { const real d = Dinv[row]; rv = (rv * d) * cinv; a = (a * d) * cinv; bq = (bq * d) * cinv; cm = (cm * d) * cinv; }
      a_rd = amax(a_rd, rv); a_md = amax(a_md, a); a_md = amax(a_md, bq); a_md = amax(a_md, cm);
      xpx += px * xv; qx += qv * xv; });
    rd = bmax<BS>(a_rd, red); md = bmax<BS>(a_md, red);
    xpx = bsum<BS>(xpx, red); qx = bsum<BS>(qx, red);
    cost = (unscale ? cinv : R(1.0)) * (R(0.5) * xpx + qx);
    __syncthreads();
  };

  // ---- admm_z!: w_prev = w ; s = Pi(w_s)  (solver.jl:151-152) ----
  auto admm_z = [&]() {
    for (int i = tid; i < n + m; i += BS) {
      const real v = w[i]; w_prev[i] = v;
      if (i >= n) s[i - n] = proj_simple(v, D.meta[i - n], bl, bu);
    }
    __syncthreads();
    for (int cI = wv; cI < D.nsoc; cI += BS / 64) {                       // SecondOrderCone (convexset.jl:100-114)
      real* x = s + D.soc_off[cI]; const int d = D.soc_dim[cI];
      if (d == 0) continue;
      const real t = x[0];
      real a = 0.0;
      for (int i = 1 + lane; i < d; i += 64) { const real v = x[i]; a += v * v; }
      const real nx = sqrt(wave_sum(a));
      if (nx <= t) {
      } else if (nx <= -t) { for (int i = lane; i < d; i += 64) x[i] = 0.0; }
      else { const real f = (nx + t) / (R(2.0) * nx); for (int i = 1 + lane; i < d; i += 64) x[i] = f * x[i]; if (lane == 0) x[0] = (nx + t) / R(2.0); }
    }
    __syncthreads();
    if constexpr (PSD) {                                                  // PsdCone / PsdConeTriangle, side <= 16 (convexset.jl:303-321, 402-412)
      batch_project_psd(D, s, ops.psd_ws, wv, lane);
      batch_project_cone3<BS>(D, s);                                      // ExponentialCone / PowerCone / duals (disjoint rows)
      __syncthreads();
      if (D.nmid > 0) {                                                   // ... and side 17 .. 64: the whole workgroup, one cone after the other
        __shared__ int any_rot_s;
        batch_project_psd_mid<BS>(D, k, s, ops.psd_ws, red, &any_rot_s);
      }
    }
  };

  // ---- the accelerator's state: block-uniform registers, kept in D.aa[k] between launches ----
  AaRegs S;
  AaMem M;
  const int N = n + m;
  if constexpr (AA) { M = aa_mem_of(D, k); aa_load(S, M.aa); }
  auto each = [&](auto&& fn) { for (int e = tid; e < N; e += BS) fn(e, w[e], w_prev[e]); };

  if (do_init) {                                                          // solver.jl:137-138
    solve_and_update();
  }
  long long it = ctl->iter;
  while (it < iter_target && it + S.sg < P.max_iter) {
    ++it;
    if constexpr (AA) {
      aa_pre<BS>(S, M, P, it, each, red);
      // delta_y of the certificates at the first non-accelerated iteration after a flagged one (solver.jl:145-148)
      if (S.inf_due && !S.success) { for (int i = tid; i < m; i += BS) D.inf_dy[om + i] = rho[i] * (w_prev[n + i] - s[i]); }
    }
    admm_z();
    // ---- apply_rho_adaptation_rules! (solver.jl:242-282, parameters.jl:53-92); with an accelerator at the next non-accelerated iteration ----
    bool do_rho = P.adaptive_rho && P.adaptive_rho_interval > 0 && (it % P.adaptive_rho_interval) == 0 &&
                  (long long)(ctl->n_rho_updates - 1) < P.max_adaptions;
    if constexpr (AA) {
      if (do_rho) S.rho_due = 1;
      do_rho = S.rho_due && !S.success;
      if (do_rho) S.rho_due = 0;
    }
    if (do_rho) {
      residuals(false);
      const real rpn = rp / (mp + R(1e-10)), rdn = rd / (md + R(1e-10));
      const real rho0 = ctl->rho;
      real nr = rho0 * sqrt(rpn / (rdn + R(1e-10)));
      nr = clamp_keep_nan(nr, P.rho_min, P.rho_max);
      const bool adapt = (nr > P.adapt_tol * rho0) || (nr < (R(1.0) / P.adapt_tol) * rho0);
      __syncthreads();
      if (adapt) {
        for (int i = tid; i < m; i += BS) {
          const int cc = cls[i]; real rv = nr;
          if (cc == 1) rv = rv * P.rho_eq; else if (cc == 2) rv = P.rho_min;
          rho[i] = rv;
          w[n + i] = (R(1.0) / rv) * mu[i] + s[i];
        }
        if (tid == 0) {
          ctl->rho = nr;
          const int ku = ctl->n_rho_updates;
          if (ku < COSMO_HIP_MAX_RHO_UPDATES) ctl->rho_updates[ku] = nr;
          ctl->n_rho_updates = ku + 1;
        }
        if constexpr (AA) { S.iter = 0; S.init = 1; }                     // CA.restart! (solver.jl:272-275): the operator changed
      }
      __syncthreads();
    }
    solve_and_update();
    // ---- acceleration_post! (accelerator_interface.jl:85-116): safeguarding ----
    if constexpr (AA) {
      if (S.active && S.success) {
        if (P.aa_safeguard) {
          if (aa_declined<BS>(S, M, P, each, red)) {                      // back to the last non-accelerated point, one plain ADMM step
            aa_reset(M, each);
            __syncthreads();
            admm_z();
            solve_and_update();
            S.sg += 1; S.n_decl += 1;
          } else S.n_ok += 1;
        }
        S.n_acc += 1;
      }
    }
    // ---- check_termination! (solver.jl:306-321) ----
    if ((it % P.check_termination) == 0 || it == 1) {
      residuals(P.unscale != 0);
      int st = 0;
      if (fabs(cost) > R(1e20)) st = COSMO_HIP_UNSOLVED;
      else if (rp < P.eps_abs + P.eps_rel * mp && rd < P.eps_abs + P.eps_rel * md &&
               ((P.obj_true != P.obj_true) || fabs(P.obj_true - cost) <= P.obj_true_tol)) st = COSMO_HIP_SOLVED;   // has_converged (residuals.jl:131-139)
      if constexpr (AA) {                                                 // check_activation!(ws, ::AccuracyActivation, r) (accelerator_interface.jl:38-46)
        if (st == 0 && !S.active && P.aa_start_acc >= R(0.0) &&
            rp < P.aa_start_acc + P.aa_start_acc * mp && rd < P.aa_start_acc + P.aa_start_acc * md) S.active = 1;
      }
      if (tid == 0) { ctl->cost = cost; ctl->r_prim = rp; ctl->r_dual = rd; ctl->max_norm_prim = mp; ctl->max_norm_dual = md; ctl->status = st; }
      __syncthreads();
      if (st != 0) break;
    }
    if constexpr (AA) {                                                   // solver.jl:326-349: the certificates wait for a non-accelerated iteration
      if (P.check_inf > 0 && (it % P.check_inf) == 0) S.inf_due = 1;
      else if (S.inf_due && !S.success) { S.inf_due = 0; S.need_inf = 1; break; }   // the host runs k_batch_inf_check on this problem, then relaunches
    }
  }
  if (tid == 0) ctl->iter = it;
  // iter (+ safeguarding_iter) == max_iter: calculate_result_info! and Max_iter_reached, overriding a status decided in that iteration
  // (solver.jl:173-176, reference quirk kept)
  if (it + S.sg >= P.max_iter) {
    __syncthreads();
    residuals(P.unscale != 0);
    if (tid == 0) { ctl->r_prim = rp; ctl->r_dual = rd; ctl->max_norm_prim = mp; ctl->max_norm_dual = md; ctl->status = COSMO_HIP_MAX_ITER_REACHED; }
  }
  if constexpr (AA) { if (tid == 0) aa_store(S, M.aa); }
  // recover_mu! (solver.jl:167)
  __syncthreads();
  for (int i = tid; i < m; i += BS) mu[i] = rho[i] * (w_prev[n + i] - s[i]);
}

template <bool PSD, bool AA>
__global__ __launch_bounds__(COSMO_BS) void k_batch_admm(BatchDev D, BParams P, long long iter_target, int do_init) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  __shared__ __attribute__((aligned(16))) unsigned char psd_ws[PSD ? (COSMO_BS / 64) * PSD16_WS_STRIDE : 16];
  const int k = blockIdx.x;
  if (D.ctl[k].status != 0) return;
  StreamOps ops;
  ops.A = bview(D.A, k); ops.AT = bview(D.AT, k); ops.PT = bview(D.PT, k); ops.lds = lds; ops.red = red; ops.psd_ws = psd_ws;
  batch_admm_body<COSMO_BS, PSD, AA>(D, P, iter_target, do_init, ops, red, k);
}

// MERGED launch over one-problem batches of DIFFERENT structure (batch_multi_optimize): the streaming form takes any size and reads everything it needs
// about its problem from the descriptor -- one workgroup per batch, one launch for all of them
template <bool PSD>
__global__ __launch_bounds__(COSMO_BS) void k_batch_admm_multi(const BatchDev* __restrict__ Ds, BParams P, long long iter_target, int do_init) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  __shared__ __attribute__((aligned(16))) unsigned char psd_ws[PSD ? (COSMO_BS / 64) * PSD16_WS_STRIDE : 16];
  const BatchDev& D = Ds[blockIdx.x];
  if (D.ctl[0].status != 0) return;
  StreamOps ops;
  ops.A = bview(D.A, 0); ops.AT = bview(D.AT, 0); ops.PT = bview(D.PT, 0); ops.lds = lds; ops.red = red; ops.psd_ws = psd_ws;
  batch_admm_body<COSMO_BS, PSD, false>(D, P, iter_target, do_init, ops, red, 0);
}

// LDS-resident variant: `img` holds one image of `img_stride` bytes per problem (header + arrays, see build_lds_images)
template <int BS, bool PSD, bool AA>
__global__ __launch_bounds__(BS) void k_batch_admm_lds(BatchDev D, BParams P, long long iter_target, int do_init,
                                                       const unsigned char* __restrict__ img, long long img_stride) {
  extern __shared__ real dyn_lds[];
  const int k = blockIdx.x;
  if (D.ctl[k].status != 0) return;
  const unsigned char* src = img + (long long)k * img_stride;
  const LdsHdr hd = *reinterpret_cast<const LdsHdr*>(src);
  {
    const real* s8 = reinterpret_cast<const real*>(src);
    const int nd = hd.bytes / (int)sizeof(real);
    for (int i = threadIdx.x; i < nd; i += BS) dyn_lds[i] = s8[i];
  }
  unsigned char* base = reinterpret_cast<unsigned char*>(dyn_lds);
  LdsOps<BS> ops;
  ops.Aval = reinterpret_cast<const real*>(base + hd.oAval); ops.Pval = reinterpret_cast<const real*>(base + hd.oPval);
  ops.Arp = reinterpret_cast<const unsigned short*>(base + hd.oArp); ops.Acol = reinterpret_cast<const unsigned short*>(base + hd.oAcol);
  ops.Trp = reinterpret_cast<const unsigned short*>(base + hd.oTrp); ops.Tpr = reinterpret_cast<const uint32_t*>(base + hd.oTpr);
  ops.Prp = reinterpret_cast<const unsigned short*>(base + hd.oPrp); ops.Pcol = reinterpret_cast<const unsigned short*>(base + hd.oPcol);
  ops.rbA = reinterpret_cast<const int4*>(base + hd.oRbA); ops.rbAT = reinterpret_cast<const int4*>(base + hd.oRbAT);
  ops.rbPT = reinterpret_cast<const int4*>(base + hd.oRbPT);
  ops.nbA = hd.nbA; ops.nbAT = hd.nbAT; ops.nbPT = hd.nbPT;
  ops.Aord = hd.oAord ? reinterpret_cast<const unsigned short*>(base + hd.oAord) : nullptr;
  ops.Tord = hd.oTord ? reinterpret_cast<const unsigned short*>(base + hd.oTord) : nullptr;
  real* wsp = reinterpret_cast<real*>(base + img_stride);            // workspace behind the image
  ops.xv = wsp; ops.tv = wsp + D.n; ops.red = wsp + D.n + D.m; ops.n = D.n;
  ops.psd_ws = base + ((img_stride + (long long)sizeof(real) * (D.n + D.m + 2 * (BS / 64)) + 15) / 16) * 16;     // behind the reduction slots (build_lds_images sizes it)
  __syncthreads();
  batch_admm_body<BS, PSD, AA>(D, P, iter_target, do_init, ops, ops.red, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident variant of the LDS kernel for problems with n <= JN*BS and m <= JM*BS: element i of every iterate
// vector is OWNED by thread i % BS and lives in that thread's registers (slot i / BS) for the whole launch; row r of every
// sparse product is computed by its owner, which finds everything row-indexed (rho, ls_s, s, w, rhs, r, c ...) in its own
// registers.  The only shared data are the two gather vectors (LDS) and the matrices (LDS).  Global memory is touched when
// the launch starts (load state), ends (store state) and at the termination checks (scaling vectors).  The arithmetic per
// element and the left-to-right row sums are those of batch_admm_body; block reductions add the per-thread partials of the
// BS-strided ownership (fixed order, deterministic).
// ---------------------------------------------------------------------------------------------------------------------
#ifdef COSMO_BATCH_TIMING
__device__ long long g_bt[8];     // lab instrumentation: shader-clock cycles of workgroup 0 per phase (not built by default)
#define BT_BEGIN() long long bt__ = clock64()
#define BT_END(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_bt[slot] += clock64() - bt__; } } while (0)
extern "C" void cosmo_dbg_batch_timing(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt), sizeof(long long) * 8); }
#else
#define BT_BEGIN()
#define BT_END(slot)
#endif

// SLICED: the image is the sliced one (row_sliced above; build_lds_images decides per batch) -- the gathered vectors and the reduction slots sit IN FRONT of
// the image at fixed LDS addresses (xv at 0, tv at 8 JN BS, the block-sum slots behind it), the image follows
// LONG (round 6): the batch has rows / columns of A with >= LONG_ROW entries; the Krylov passes of the row-major sorted form hand them to the whole wave
// (row_long_or_pipe3).  A separate instantiation: the kernels of the batches without such rows are the same code as before.
template <int BS, int JN, int JM, bool PSD, bool AA, bool SLICED = false, bool LONG = false>
__global__ __launch_bounds__(BS) void k_batch_admm_reg(BatchDev D, BParams P, long long iter_target, int do_init,
                                                       const unsigned char* __restrict__ img, long long img_stride) {
  extern __shared__ real dyn_lds[];
  static_assert(!SLICED || (JN == 1 && BS == 512 && !REAL_IS_FLOAT), "the sliced image exists for the <512, 1, 2> double-precision instantiations");
  static_assert(!LONG || !SLICED, "the cooperative long-row passes exist for the row-major forms (<512, 1, 2> sorted, <512, 2, 4> index order)");
  constexpr int WS0 = SLICED ? (int)sizeof(real) * (JN * BS + JM * BS + 2 * (BS / 64)) : 0;     // bytes in front of the image
  constexpr int TVOFF = (int)sizeof(real) * JN * BS;                                              // LDS address of tv (sliced form)
  (void)TVOFF;
  const int k = blockIdx.x;
  BCtl* ctl = D.ctl + k;
  if (ctl->status != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wvid = tid >> 6;
  const int n = D.n, m = D.m;
  const unsigned char* src = img + (long long)k * img_stride;
  const LdsHdr hd = *reinterpret_cast<const LdsHdr*>(src);
  {
    const real* s8 = reinterpret_cast<const real*>(src);
    const int nd = hd.bytes / (int)sizeof(real);
    for (int i = tid; i < nd; i += BS) dyn_lds[WS0 / (int)sizeof(real) + i] = s8[i];
  }
  unsigned char* base = reinterpret_cast<unsigned char*>(dyn_lds) + WS0;
  const real* Aval = reinterpret_cast<const real*>(base + hd.oAval);
  const real* Pval = reinterpret_cast<const real*>(base + hd.oPval);
  const unsigned short* Arp = reinterpret_cast<const unsigned short*>(base + hd.oArp);
  const unsigned short* Acol = reinterpret_cast<const unsigned short*>(base + hd.oAcol);
  const unsigned short* Trp = reinterpret_cast<const unsigned short*>(base + hd.oTrp);
  const uint32_t* Tpr = reinterpret_cast<const uint32_t*>(base + hd.oTpr);
  const unsigned short* Prp = reinterpret_cast<const unsigned short*>(base + hd.oPrp);
  const unsigned short* Pcol = reinterpret_cast<const unsigned short*>(base + hd.oPcol);
  real* wsp = SLICED ? dyn_lds : reinterpret_cast<real*>(base + img_stride);
  real* xv = wsp;                 // n : vector gathered by the A / P products
  real* tv = SLICED ? wsp + JN * BS : wsp + n;             // m : vector gathered by the A' products; staging of s for the SOC projection
  real* red = SLICED ? wsp + JN * BS + JM * BS : wsp + n + m;        // 2 x (BS / 64) reduction slots (the second set: double-buffered block sums of the Krylov loop, bsum_db)
  unsigned char* psd_ws = SLICED ? reinterpret_cast<unsigned char*>(dyn_lds) + ((WS0 + img_stride + 15) / 16) * 16
                                 : base + ((img_stride + (long long)sizeof(real) * (n + m + 2 * (BS / 64)) + 15) / 16) * 16;   // wave workspaces of the small PSD cones
  (void)psd_ws;

  const long long on = (long long)k * n, om = (long long)k * m, onm = (long long)k * (n + m);
  // OWNERSHIP of the n-vectors (round 5): slot j of thread t holds element ct[j] -- the column of the length-sorted compute assignment of the
  // Krylov loop's column pass (identity without the table: COSMO_HIP_BATCH_SORTED=0 and the <512, 2, 4> instantiation).  Until round 4 the owner of
  // element i was thread i mod 512 and the computing thread handed c = P u + sigma u + A' tmp over through LDS (two barriers and two LDS accesses
  // per Krylov iteration); now owner == computer, at the price of bit-identity with the owner-computes FORM: the block sums add the same per-element
  // terms in another thread order (run-to-run deterministic; parity = the tight-CG fixture at 1e-7, tests/test_gpu_batch.py).
  constexpr bool SORTED = (JN == 1);
  int ct[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int i = tid + BS * j;
    ct[j] = (SORTED && D.permT) ? D.permT[(long long)k * (JN * BS) + j * BS + tid] : (i < n ? i : -1);
  }
  // (the index-order instantiation <512, 2, 4> recomputes its elements from the thread index instead of holding them: it has no registers to spare)
  auto OWN = [&](int j) -> int { if constexpr (SORTED) return ct[j]; else { const int i = tid + BS * j; return i < n ? i : -1; } };
  // ---- load the persistent state and the per-element constants into registers -----------------------------------------
  real wx[JN], wpx[JN], qv[JN], xtl[JN], lsx[JN], rhsv[JN], rv[JN], cv[JN];
  real wsv[JM], wps[JM], sv[JM], rhov[JM], lss[JM], bv[JM], blv[JM], buv[JM];
  uint32_t metav[JM];
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int i = OWN(j);
    const bool ok = i >= 0;
    wx[j] = ok ? D.w[onm + i] : 0.0; wpx[j] = ok ? D.w_prev[onm + i] : 0.0; qv[j] = ok ? D.q[on + i] : 0.0; xtl[j] = ok ? D.x_tl[on + i] : 0.0;
    lsx[j] = 0.0; rhsv[j] = 0.0; rv[j] = 0.0; cv[j] = 0.0;
  }
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    const int i = tid + BS * j;
    const bool ok = i < m;
    wsv[j] = ok ? D.w[onm + n + i] : 0.0; wps[j] = ok ? D.w_prev[onm + n + i] : 0.0; sv[j] = ok ? D.s[om + i] : 0.0;
    rhov[j] = ok ? D.rho[om + i] : 1.0; bv[j] = ok ? D.b[om + i] : 0.0; lss[j] = 0.0;
    metav[j] = ok ? D.meta[i] : 0u;
    blv[j] = 0.0; buv[j] = 0.0;
    if (ok && (metav[j] & 3u) == 3u) { const uint32_t jb = metav[j] >> 2; blv[j] = D.box_l[(long long)k * D.nbox + jb]; buv[j] = D.box_u[(long long)k * D.nbox + jb]; }
  }
  const int* cls = D.rho_cls + om;
  const real cinv = D.cinv[k];
  // compute assignment of the Krylov loop's sparse passes (rows / columns sorted by length; identity without the tables) and rho of the
  // rows this thread computes (kept in step with rhov by the same formula at every adaptation)
  // (the <512, 2, 4> instantiation is at its 256 registers already: it keeps the owner-computes form)
  int ra[JM];
  real rhoc[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    const int i = tid + BS * j;
    ra[j] = (SORTED && D.permA) ? D.permA[(long long)k * (JM * BS) + j * BS + tid] : (i < m ? i : -1);
    rhoc[j] = (SORTED && ra[j] >= 0) ? D.rho[om + ra[j]] : R(1.0);
  }
  // POSITIONS of the gathered vectors: entry i of an n-vector sits at xv[posN[i]], entry r of an m-vector at tv[posM[r]] whenever a sparse
  // pass gathers from them (the image's column / row indices are stored as positions).  The host chooses the positions so that the rows
  // a wave-step works on spread over the LDS banks (build_lds_images); without the tables positions are indices.
  int pn[JN], pm[JM], pnc[JN], pmc[JM];
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    pnc[j] = (SORTED && D.posN && OWN(j) >= 0) ? D.posN[on + OWN(j)] : OWN(j);
    pn[j] = pnc[j];                                    // owner == computer
  }
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    const int i = tid + BS * j;
    pm[j] = (SORTED && D.posM && i < m) ? D.posM[om + i] : i;
    pmc[j] = (SORTED && D.posM && ra[j] >= 0) ? D.posM[om + ra[j]] : ra[j];
  }
  long long solves = ctl->solves, kkt_total = ctl->kkt_iters_total;
  int n_rho = ctl->n_rho_updates;
  real rho_s = ctl->rho;
  // the cones of this wave (cone c belongs to wave c % (BS/64)): offsets / dimensions in registers when they fit
  constexpr int JS = 8;
  const bool soc_in_regs = D.nsoc <= JS * (BS / 64);
  int soc_o[JS], soc_d[JS];
#pragma unroll
  for (int t = 0; t < JS; ++t) {
    const int cI = wvid + (BS / 64) * t;
    const bool ok = soc_in_regs && cI < D.nsoc;
    soc_o[t] = ok ? D.soc_off[cI] : 0; soc_d[t] = ok ? D.soc_dim[cI] : 0;
  }
  real tol_next = D.tol_table[solves < D.tol_len ? solves : D.tol_len - 1];   // requested one solve ahead: its latency is hidden
  __syncthreads();

  // ---- row products on the LDS image: the owner of row r adds its products left to right -----------------------------
  // (measured on config 3: a sparse pass costs ~25 cycles per wave step of three LDS reads, set by LDS issue rate and by the
  //  spread of row lengths inside a wave, not by latency -- manual unrolling, 16 waves instead of 8 and a sliced-JDS layout
  //  with conflict-free value / index reads were all tried and were the same speed or slower)
  // (four nonzeros per trip with their loads issued together -- tried again in round 3 on top of the length-sorted assignment below, where
  //  the rows of a wave have similar lengths: 3608 vs 4129 batch-it/s on config 3, slower as in round 2)
  // Software-pipelined by one nonzero (round 4): the index (and value) loads of nonzero t + 1 are issued together with the gather of nonzero t, so
  // that a trip costs ONE dependent LDS round trip instead of two (index -> gather).  Same left-to-right sums: bit-identical to the plain loops
  // `for (t = a; t < b; ++t) s1 += Aval[t] * xv[Acol[t]]` (iterate hashes equal on all 1024 problems of config 3), 4100 -> 4367 batch-it/s
  // (7.35 -> 6.95 us per Krylov iteration of the slowest problem; profiles/r04_batch_pipe_lab.txt).
  // Two-stage software pipeline (round 5): the gathered operand of nonzero t is REQUESTED one trip before it is multiplied -- trip t issues the
  // gather of t + 1 (its index arrived a trip ago) and the index / value loads of t + 2, then consumes the gather of t.  With the one-stage form
  // (index -> gather in consecutive trips, gather -> product inside a trip) every trip exposed one LDS round trip: the ISA of the loop was
  // gather, wait, multiply.  Same products added in the same order; the last trips re-load the row's last element instead of branching (clamped
  // index, value unused).  COSMO_BATCH_PIPE2=0 at compile time restores the one-stage loops.
#ifndef COSMO_BATCH_PIPE2
#define COSMO_BATCH_PIPE2 1
#endif
  auto rowA_b = [&](int t, const int b2) -> real {
    real s1 = 0.0;
    if (t < b2) {
#if COSMO_BATCH_PIPE2
      const int last = b2 - 1;
      real v = Aval[t]; int c = Acol[t];
      int t1 = (t + 1 < b2) ? t + 1 : last;
      real vn = Aval[t1]; int cn = Acol[t1];
      real g = xv[c];
      for (++t; t < b2; ++t) {
        const real gn = xv[cn];                                    // gather of nonzero t (address known since the last trip)
        const int t2 = (t + 1 < b2) ? t + 1 : last;
        const real v2 = Aval[t2]; const int c2 = Acol[t2];         // loads of nonzero t + 1
        s1 += v * g;                                               // nonzero t - 1: its gather was requested a trip ago
        v = vn; g = gn; vn = v2; cn = c2;
      }
      s1 += v * g;
#else
      real v = Aval[t]; int c = Acol[t];
      for (++t; t < b2; ++t) { const real vn = Aval[t]; const int cn = Acol[t]; s1 += v * xv[c]; v = vn; c = cn; }
      s1 += v * xv[c];
#endif
    }
    return s1 + R(0.0);
  };
  auto rowA = [&](int r) -> real {
    real s1 = 0.0;
    int t = Arp[r]; const int b2 = Arp[r + 1];
    if (t < b2) {
      real v = Aval[t]; int c = Acol[t];
      for (++t; t < b2; ++t) { const real vn = Aval[t]; const int cn = Acol[t]; s1 += v * xv[c]; v = vn; c = cn; }
      s1 += v * xv[c];
    }
    return s1 + R(0.0);
  };
  auto rowAT_b = [&](int t, const int b2) -> real {
    real s1 = 0.0;
    if (t < b2) {
#if COSMO_BATCH_PIPE2
      const int last = b2 - 1;
      uint32_t pr = Tpr[t];
      uint32_t prn = Tpr[(t + 1 < b2) ? t + 1 : last];
      real a = Aval[pr & 0xffffu], g = tv[pr >> 16];
      for (++t; t < b2; ++t) {
        const real an = Aval[prn & 0xffffu], gn = tv[prn >> 16];   // value and gathered operand of nonzero t
        const uint32_t pr2 = Tpr[(t + 1 < b2) ? t + 1 : last];     // packed pair of nonzero t + 1
        s1 += a * g;                                               // nonzero t - 1
        a = an; g = gn; prn = pr2;
      }
      s1 += a * g;
#else
      uint32_t pr = Tpr[t];
      for (++t; t < b2; ++t) { const uint32_t prn = Tpr[t]; s1 += Aval[pr & 0xffffu] * tv[pr >> 16]; pr = prn; }
      s1 += Aval[pr & 0xffffu] * tv[pr >> 16];
#endif
    }
    return s1;
  };
  auto rowAT = [&](int r) -> real {
    real s1 = 0.0;
    int t = Trp[r]; const int b2 = Trp[r + 1];
    if (t < b2) {
      uint32_t pr = Tpr[t];
      for (++t; t < b2; ++t) { const uint32_t prn = Tpr[t]; s1 += Aval[pr & 0xffffu] * tv[pr >> 16]; pr = prn; }
      s1 += Aval[pr & 0xffffu] * tv[pr >> 16];
    }
    return s1;
  };
  auto rowP_b = [&](int t, const int b2) -> real {
    real s1 = 0.0;
    if (t < b2) {
      real v = Pval[t]; int c = Pcol[t];
      for (++t; t < b2; ++t) { const real vn = Pval[t]; const int cn = Pcol[t]; s1 += v * xv[c]; v = vn; c = cn; }
      s1 += v * xv[c];
    }
    return s1;
  };
  auto rowP = [&](int r) -> real {
    real s1 = 0.0;
    int t = Prp[r]; const int b2 = Prp[r + 1];
    if (t < b2) {
      real v = Pval[t]; int c = Pcol[t];
      for (++t; t < b2; ++t) { const real vn = Pval[t]; const int cn = Pcol[t]; s1 += v * xv[c]; v = vn; c = cn; }
      s1 += v * xv[c];
    }
    return s1;
  };
  // Row bounds of the rows / the column this thread computes in EVERY Krylov iteration (constant for the launch): kept in registers, so that a row
  // starts with its first value / index loads instead of a dependent row-pointer round trip (8 LDS reads and 4 dependent round trips less per thread
  // and Krylov iteration).  The <512, 2, 4> instantiation has no registers to spare and keeps reading them.
  int ka0[SORTED ? JM : 1], ka1[SORTED ? JM : 1], kp0[SORTED ? JN : 1], kp1[SORTED ? JN : 1], kt0[SORTED ? JN : 1], kt1[SORTED ? JN : 1];
  const bool stored_sorted = SORTED && D.qposA != nullptr;
  // hand-scheduled row loops of the Krylov passes (row_pipe3 above); -DCOSMO_BATCH_HANDPIPE=0 keeps the compiled loops (lab builds)
#ifndef COSMO_BATCH_HANDPIPE
#define COSMO_BATCH_HANDPIPE 1
#endif
  constexpr bool HANDPIPE = SORTED && (COSMO_BATCH_HANDPIPE != 0);
  const uint32_t lA_val = lds_addr_of(Aval), lA_col = lds_addr_of(Acol), lT_pr = lds_addr_of(Tpr), l_xv = lds_addr_of(xv), l_tv = lds_addr_of(tv);
  if constexpr (SORTED && !SLICED) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int r = ra[j];
      const int q = stored_sorted ? (BS * j + ((j & 1) ? (BS - 1 - tid) : tid)) : r;       // build_lds_images: slot j of thread t holds sorted position 512 j + (t | 511 - t)
      ka0[j] = r >= 0 ? Arp[q] : 0; ka1[j] = r >= 0 ? Arp[q + 1] : 0;
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int c = OWN(j);
      const int q = stored_sorted ? (BS * j + tid) : c;                                   // columns: sorted position q sits in slot q / 512 of thread q % 512
      kp0[j] = c >= 0 ? Prp[q] : 0; kp1[j] = c >= 0 ? Prp[q + 1] : 0; kt0[j] = c >= 0 ? Trp[q] : 0; kt1[j] = c >= 0 ? Trp[q + 1] : 0;
    }
  }
  // sliced image: first entries / lengths of the rows and the column this thread computes, the wave maxima of the lengths (scalar trip counts)
  uint32_t sa_ip[SLICED ? JM : 1], sa_vp[SLICED ? JM : 1], st_ip[SLICED ? JN : 1];
  int sa_len[SLICED ? JM : 1], sa_L[SLICED ? JM : 1], st_len[SLICED ? JN : 1], st_L[SLICED ? JN : 1];
  real pdv[SLICED ? JN : 1];
  bool pdh[SLICED ? JN : 1];
  (void)sa_ip; (void)sa_vp; (void)st_ip; (void)sa_len; (void)sa_L; (void)st_len; (void)st_L; (void)pdv; (void)pdh;
#if !REAL_IS_FLOAT
  if constexpr (SLICED) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const uint32_t e = D.slA[((long long)k * JM + j) * BS + tid];
      sa_len[j] = (int)(e >> 16); sa_ip[j] = lA_col + 2u * (e & 0xffffu); sa_vp[j] = lA_val + 8u * (e & 0xffffu);
      sa_L[j] = wave_max_int(sa_len[j]);
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const uint32_t e = D.slT[((long long)k * JN + j) * BS + tid];
      st_len[j] = (int)(e >> 16); st_ip[j] = lT_pr + 4u * (e & 0xffffu);
      st_L[j] = wave_max_int(st_len[j]);
      const int c = OWN(j);
      pdv[j] = (D.pdiag && c >= 0) ? D.pdiag[on + c] : R(0.0);
      pdh[j] = D.pdiag && c >= 0 && D.pdiag_has[on + c] != 0;
      if (!D.pdiag) { const int q = BS * j + tid; kp0[j] = c >= 0 ? Prp[q] : 0; kp1[j] = c >= 0 ? Prp[q + 1] : 0; }
    }
  }
#endif
  // every row / column product of this kernel goes through these three: the bounds held in registers (sorted instantiation) or the row pointers
  // (the rows a thread OWNS are used outside the Krylov loop only -- three times per ADMM iteration: their position is read when needed, no registers held)
  auto rowA_own = [&](int j) -> real {
    const int i = tid + BS * j;
    if constexpr (SORTED) { const int qo = stored_sorted ? D.qposA[om + i] : i; return rowA_b(Arp[qo], Arp[qo + 1]); }
    else return rowA(i);
  };
  auto rowAT_own = [&](int j) -> real { if constexpr (SORTED) return rowAT_b(kt0[j], kt1[j]); else return rowAT(OWN(j)); };
  auto rowP_own = [&](int j) -> real { if constexpr (SORTED) return rowP_b(kp0[j], kp1[j]); else return rowP(OWN(j)); };
  // sliced image: the rows ra[] of the compute assignment (A), the owned column (A', P); vj = the gathered vector's entry of the owned column (what a
  // diagonal P held in registers multiplies: xv[position of c] is this thread's own element)
#if !REAL_IS_FLOAT
  auto A_sl = [&](int j) -> real { return row_sliced<0>(sa_ip[j], sa_vp[j], sa_len[j], sa_L[j]) + R(0.0); };
  auto T_sl = [&](int j) -> real { return rowT_sliced<TVOFF>(st_ip[j], st_len[j], st_L[j]); };
  auto P_sl = [&](int j, const real vj) -> real {
    if (D.pdiag) return pdh[j] ? R(0.0) + pdv[j] * vj : R(0.0);
    return rowP_b(kp0[j], kp1[j]);
  };
#else
  auto A_sl = [&](int) -> real { return R(0.0); };
  auto T_sl = [&](int) -> real { return R(0.0); };
  auto P_sl = [&](int, const real) -> real { return R(0.0); };
#endif
  // (A x)_i for the rows a thread OWNS, sliced form: the computing threads hand their rows over through tv (xv holds x, published and fenced by the caller;
  // tv is free: nobody reads it between the caller's last barrier and here).  Ends with a barrier after which tv may be overwritten.
  auto Ax_to_owners = [&](real* ax) {
    real t_[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) t_[j] = (ra[j] >= 0) ? A_sl(j) : R(0.0);
#pragma unroll
    for (int j = 0; j < JM; ++j) if (ra[j] >= 0) tv[pmc[j]] = t_[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; ax[j] = (i < m) ? tv[pm[j]] : R(0.0); }
    __syncthreads();
  };
  (void)Ax_to_owners;
  auto colT = [&](int j) -> real { if constexpr (SLICED) return T_sl(j); else return rowAT_own(j); };
  auto colP = [&](int j, const real vj) -> real { if constexpr (SLICED) return P_sl(j, vj); else { (void)vj; return rowP_own(j); } };
  // LONG instantiations: which lanes hand their row to the wave -- per slot, for the rows of the Krylov A pass (mkA: the sorted assignment), the rows a
  // thread owns (mkO: index order; the same rows in the index-order form) and the columns (mkT)
  unsigned long long mkA[LONG ? JM : 1], mkO[LONG ? JM : 1], mkT[LONG ? JN : 1];
  (void)mkA; (void)mkO; (void)mkT;
  if constexpr (LONG) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      int lo = 0;
      if (i < m) { int qo = i; if constexpr (SORTED) { if (stored_sorted) qo = D.qposA[om + i]; } lo = (int)Arp[qo + 1] - (int)Arp[qo]; }
      mkO[j] = long_row_mask(lo);
      if constexpr (SORTED) mkA[j] = long_row_mask(ka1[j] - ka0[j]); else mkA[j] = mkO[j];
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      int lt;
      if constexpr (SORTED) lt = kt1[j] - kt0[j];
      else { const int c = OWN(j); lt = c >= 0 ? (int)Trp[c + 1] - (int)Trp[c] : 0; }
      mkT[j] = long_row_mask(lt);
    }
  }
  // LONG instantiations: the passes OUTSIDE the Krylov loop -- (A x)_i for the rows a thread owns, (A' y)_c for its columns -- with the rows of >= LONG_ROW
  // entries walked by the whole wave (called by every lane: no per-lane condition around them)
  auto ownA_all = [&](real* out) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      int a0 = 0, len = 0;
      if (i < m) { int qo = i; if constexpr (SORTED) { if (stored_sorted) qo = D.qposA[om + i]; } a0 = (int)Arp[qo]; len = (int)Arp[qo + 1] - a0; }
      out[j] = row_long_or_pipe3<false>(lA_col + 2u * (uint32_t)a0, lA_val + ((uint32_t)a0 << RSH), l_xv, len, mkO[j]) + R(0.0);
    }
  };
  auto colT_all = [&](real* out) {
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      int t0, t1;
      if constexpr (SORTED) { t0 = kt0[j]; t1 = kt1[j]; }
      else { const int c = OWN(j); t0 = c >= 0 ? (int)Trp[c] : 0; t1 = c >= 0 ? (int)Trp[c + 1] : 0; }
      out[j] = row_long_or_pipe3<true>(lT_pr + 4u * (uint32_t)t0, lA_val, l_tv, t1 - t0, mkT[j]);
    }
  };
  (void)ownA_all; (void)colT_all;

  // ---- admm_x! + admm_w! (solver.jl:32-65) with the CG reduced solve (kktsolver_indirect.jl:36-88) -------------------
  auto solve_and_update = [&]() {
#pragma unroll
    for (int j = 0; j < JN; ++j) lsx[j] = P.sigma * wx[j] - qv[j];
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      const real v = (bv[j] - R(2.0) * sv[j]) + wsv[j];
      lss[j] = v;
      if (i < m) tv[pm[j]] = rhov[j] * v;                                // y2 = rho .* ls_s
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) { const int i = OWN(j); if (i >= 0) xv[pn[j]] = xtl[j]; }
    __syncthreads();
    real acc = 0.0;
    real tall[JN];
    (void)tall;
    if constexpr (LONG) colT_all(tall);
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int i = OWN(j);
      if (i >= 0) { real ct_; if constexpr (LONG) ct_ = tall[j]; else ct_ = colT(j); const real v = (ct_ + R(0.0)) + lsx[j]; rhsv[j] = v; acc += v * v; }
    }
    const real bb = bsum<BS>(acc, red);                                  // (its barriers also order tv reads before the writes below)
    real tmpv[JM];
    if constexpr (SLICED) {
      // rho .* (A x_tl) by the computing threads, straight to its consumers (the same values the owners would have written)
#pragma unroll
      for (int j = 0; j < JM; ++j) tmpv[j] = (ra[j] >= 0) ? A_sl(j) * rhoc[j] : 0.0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < JM; ++j) if (ra[j] >= 0) tv[pmc[j]] = tmpv[j];
      __syncthreads();
    } else {
    if constexpr (LONG) {
      ownA_all(tmpv);
#pragma unroll
      for (int j = 0; j < JM; ++j) tmpv[j] = tmpv[j] * rhov[j];           // (rows beyond m: 0 x 1)
    } else {
#pragma unroll
    for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; tmpv[j] = (i < m) ? rowA_own(j) * rhov[j] : 0.0; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; if (i < m) tv[pm[j]] = tmpv[j]; }
    __syncthreads();
    }
    acc = 0.0;
    if constexpr (LONG) colT_all(tall);
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int i = OWN(j);
      if (i >= 0) { real ct_; if constexpr (LONG) ct_ = tall[j]; else ct_ = colT(j); const real cj = colP(j, xtl[j]) + (P.sigma * xtl[j] + ct_); const real rj = rhsv[j] - cj; rv[j] = rj; acc += rj * rj; }
    }
    real rr = bsum<BS>(acc, red);
    const real tol_k = tol_next;
    tol_next = D.tol_table[solves + 1 < D.tol_len ? solves + 1 : D.tol_len - 1];
    const real tol = tol_k / sqrt(bb);
    real res = sqrt(rr), prev = 1.0;
    int kk = 0;
    real uv[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) uv[j] = 0.0;
    int ph = 0;                                                          // bsum_db: which of the two slot sets the next block sum writes
    if constexpr (SORTED) __syncthreads();                               // (the block sums above read `red`: see bsum_db)
    while (kk < n && !(res <= tol)) {                                    // cg! (IterativeSolvers v0.9), maxiter = n
      { BT_BEGIN();
      const real beta = (res * res) / (prev * prev);
#pragma unroll
      for (int j = 0; j < JN; ++j) { const int i = OWN(j); uv[j] = rv[j] + beta * ((kk == 0) ? R(0.0) : uv[j]); if (i >= 0) xv[pn[j]] = uv[j]; }
      __syncthreads();
      BT_END(6); }
      // The two sparse passes of a Krylov iteration are bound by the CU's LDS pipe, and a wave issues as many steps as its LONGEST row.
      // So thread t COMPUTES the rows ra[] / the column ct[] of the length-sorted assignment (rows of similar length share a wave-step:
      // about a third fewer LDS instructions on BASELINE config 3).  rho .* (A u) goes to its consumers through tv anyway; the column ct[] is
      // also the element of the n-vectors this thread OWNS (round 5), so c = P u + sigma u + A' tmp stays in its registers -- until round 4 it
      // went back to an index-order owner through xv: two more barriers and two more LDS accesses per Krylov iteration.  Every row sum is the
      // same left-to-right sum as in every other kernel form; the block sums add their terms in the order of this ownership.
      if constexpr (SLICED) {
      { BT_BEGIN();
#pragma unroll
      for (int j = 0; j < JM; ++j) tmpv[j] = (ra[j] >= 0) ? A_sl(j) * rhoc[j] : 0.0;
#pragma unroll
      for (int j = 0; j < JM; ++j) if (ra[j] >= 0) tv[pmc[j]] = tmpv[j];
      __syncthreads();
      BT_END(0); }
      } else if (SORTED) {
      { BT_BEGIN();
#pragma unroll
      for (int j = 0; j < JM; ++j)
        if constexpr (LONG) tmpv[j] = (row_long_or_pipe3<false>(lA_col + 2u * (uint32_t)ka0[j], lA_val + ((uint32_t)ka0[j] << RSH), l_xv, ka1[j] - ka0[j], mkA[j]) + R(0.0)) * rhoc[j];   // (no row: ka0 = ka1 = 0, rhoc = 1)
        else
        tmpv[j] = (ra[j] >= 0) ? (HANDPIPE ? (row_pipe3<false>(lA_col + 2u * (uint32_t)ka0[j], lA_val + ((uint32_t)ka0[j] << RSH), l_xv, ka1[j] - ka0[j]) + R(0.0))
                                           : rowA_b(ka0[j], ka1[j])) * rhoc[j] : 0.0;
#pragma unroll
      for (int j = 0; j < JM; ++j) if (ra[j] >= 0) tv[pmc[j]] = tmpv[j];   // tv was last read before the previous barrier pair
      __syncthreads();
      BT_END(0); }
      } else {
      if constexpr (LONG) {                                                // index-order form <512, 2, 4>: the bounds are read per iteration (no registers to hold them)
#pragma unroll
        for (int j = 0; j < JM; ++j) {
          const int i = tid + BS * j;
          const int a0 = (i < m) ? (int)Arp[i] : 0, a1 = (i < m) ? (int)Arp[i + 1] : 0;
          tmpv[j] = row_long_or_pipe3<false>(lA_col + 2u * (uint32_t)a0, lA_val + ((uint32_t)a0 << RSH), l_xv, a1 - a0, mkO[j]) * rhov[j];
        }
      } else {
#pragma unroll
      for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; tmpv[j] = (i < m) ? rowA(i) * rhov[j] : 0.0; }
      }
#pragma unroll
      for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; if (i < m) tv[i] = tmpv[j]; }
      __syncthreads();
      }
      acc = 0.0;
      { BT_BEGIN();
      real tlong[JN];
      (void)tlong;
      if constexpr (LONG) {                                                // (all lanes: a lane without a column passes an empty range)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          int t0, t1;
          if constexpr (SORTED) { t0 = kt0[j]; t1 = kt1[j]; }
          else { const int c = OWN(j); t0 = c >= 0 ? (int)Trp[c] : 0; t1 = c >= 0 ? (int)Trp[c + 1] : 0; }
          tlong[j] = row_long_or_pipe3<true>(lT_pr + 4u * (uint32_t)t0, lA_val, l_tv, t1 - t0, mkT[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {                                       // the column this thread owns AND computes
        const int c = OWN(j);
        if (c >= 0) {
          const real vj = uv[j];
          real cj;
          if constexpr (SLICED) cj = colP(j, vj) + (P.sigma * vj + colT(j));
          else if constexpr (LONG && SORTED) cj = rowP_b(kp0[j], kp1[j]) + (P.sigma * vj + tlong[j]);
          else if constexpr (LONG) cj = rowP(c) + (P.sigma * vj + tlong[j]);
          else if constexpr (SORTED) cj = rowP_b(kp0[j], kp1[j]) + (P.sigma * vj + (HANDPIPE ? row_pipe3<true>(lT_pr + 4u * (uint32_t)kt0[j], lA_val, l_tv, kt1[j] - kt0[j])
                                                                                                   : rowAT_b(kt0[j], kt1[j])));
          else cj = rowP(c) + (P.sigma * vj + rowAT(c));
          cv[j] = cj; acc += vj * cj;
        }
      }
      BT_END(1); }
      BT_BEGIN();
      real uc;
      if constexpr (SORTED) uc = bsum_db<BS>(acc, red, ph); else uc = bsum<BS>(acc, red);
      BT_END(2);
#ifdef COSMO_BATCH_TIMING
      if (blockIdx.x == 0 && tid == 0) g_bt[3] += 1;
#endif
      { BT_BEGIN();
      const real a = (res * res) / uc;
      acc = 0.0;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int i = OWN(j);
        if (i >= 0) { xtl[j] = xtl[j] + a * uv[j]; const real ri = rv[j] - a * cv[j]; rv[j] = ri; acc += ri * ri; }
      }
      if constexpr (SORTED) rr = bsum_db<BS>(acc, red, ph); else rr = bsum<BS>(acc, red);
      prev = res; res = sqrt(rr); ++kk;
      BT_END(7); }
    }
    // nu = rho (A x_tl - ls_s) ; s_tl ; w update
#pragma unroll
    for (int j = 0; j < JN; ++j) { const int i = OWN(j); if (i >= 0) xv[pn[j]] = xtl[j]; }
    __syncthreads();
    real axo[JM];
    if constexpr (SLICED) Ax_to_owners(axo);
    else if constexpr (LONG) ownA_all(axo);
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      if (i < m) {
        real ax_;
        if constexpr (SLICED || LONG) ax_ = axo[j]; else ax_ = rowA_own(j);
        const real rh = rhov[j]; const real nv = (ax_ - lss[j]) * rh;
        const real st = (R(2.0) * sv[j] - wsv[j]) - nv / rh;
        wsv[j] = wsv[j] + P.alpha * (st - sv[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) wx[j] = wx[j] + P.alpha * (xtl[j] - wx[j]);
    solves += 1; kkt_total += kk;
    __syncthreads();
  };

  // ---- residuals (residuals.jl:30-96,143-147); x = w_prev[1:n], mu recovered on the fly ------------------------------
  real rp, mp, rd, md, cost;
  real muv[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) muv[j] = 0.0;
  auto residuals = [&](bool unscale) {
#pragma unroll
    for (int j = 0; j < JN; ++j) { const int i = OWN(j); if (i >= 0) xv[pn[j]] = wpx[j]; }
    __syncthreads();
    real a_rp = 0.0, a_mp = 0.0;
    real axo[JM];
    if constexpr (SLICED) Ax_to_owners(axo);
    else if constexpr (LONG) ownA_all(axo);
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      if (i < m) {
        real ax;
        if constexpr (SLICED || LONG) ax = axo[j]; else ax = rowA_own(j);
        const real s0 = sv[j], b0 = bv[j];
        muv[j] = rhov[j] * (wps[j] - s0);
        tv[pm[j]] = muv[j];
        real r0 = ax + s0; r0 = r0 - b0;
        const real e = unscale ? D.Einv[om + i] : 1.0;
        if (unscale) r0 = r0 * e;
        a_rp = amax(a_rp, r0);
        a_mp = amax(a_mp, unscale ? ax * e : ax); a_mp = amax(a_mp, unscale ? s0 * e : s0); a_mp = amax(a_mp, unscale ? b0 * e : b0);
      }
    }
    rp = bmax<BS>(a_rp, red); mp = bmax<BS>(a_mp, red);
    __syncthreads();
    real a_rd = 0.0, a_md = 0.0, xpx = 0.0, qx = 0.0;
    real tall[JN];
    (void)tall;
    if constexpr (LONG) colT_all(tall);
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int i = OWN(j);
      if (i >= 0) {
        real atm; if constexpr (LONG) atm = tall[j]; else atm = colT(j);
        const real px = colP(j, wpx[j]), x0 = wpx[j], q0 = qv[j];
        real r0 = px + q0; r0 = r0 - atm;
        real a = px, bq = q0, cm = atm;
        if (unscale) { const real d = D.Dinv[on + i]; r0 = (r0 * d) * cinv; a = (a * d) * cinv; bq = (bq * d) * cinv; cm = (cm * d) * cinv; }
        a_rd = amax(a_rd, r0); a_md = amax(a_md, a); a_md = amax(a_md, bq); a_md = amax(a_md, cm);
        xpx += px * x0; qx += q0 * x0;
      }
    }
    rd = bmax<BS>(a_rd, red); md = bmax<BS>(a_md, red);
    xpx = bsum<BS>(xpx, red); qx = bsum<BS>(qx, red);
    cost = (unscale ? cinv : R(1.0)) * (R(0.5) * xpx + qx);
    __syncthreads();
  };

#ifdef COSMO_BATCH_TIMING
  const long long bt_kernel_t0 = clock64();
#endif
  // ---- admm_z!: w_prev = w ; s = Pi(w_s)  (solver.jl:151-152) ----
  auto admm_z = [&]() {
#pragma unroll
    for (int j = 0; j < JN; ++j) wpx[j] = wx[j];
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int i = tid + BS * j;
      const real v = wsv[j]; wps[j] = v;
      const uint32_t kind = metav[j] & 3u;
      real pv = v;
      if (kind == 1u) pv = 0.0;
      else if (kind == 2u) pv = (v != v) ? v : ((v > R(0.0)) ? v : R(0.0));
      else if (kind == 3u) pv = (v < blv[j]) ? blv[j] : ((v > buv[j]) ? buv[j] : v);
      sv[j] = pv;
      if ((D.nsoc > 0 || PSD) && i < m) tv[i] = pv;
    }
    if (D.nsoc > 0 || PSD) {
      __syncthreads();
      auto soc_one = [&](real* x, int d) {                              // SecondOrderCone (convexset.jl:100-114), on the LDS copy
        if (d == 0) return;
        const real t = x[0];
        real a = 0.0;
        for (int i = 1 + lane; i < d; i += 64) { const real v = x[i]; a += v * v; }
        const real nx = sqrt(wave_sum(a));
        if (nx <= t) {
        } else if (nx <= -t) { for (int i = lane; i < d; i += 64) x[i] = 0.0; }
        else { const real f = (nx + t) / (R(2.0) * nx); for (int i = 1 + lane; i < d; i += 64) x[i] = f * x[i]; if (lane == 0) x[0] = (nx + t) / R(2.0); }
      };
      if (soc_in_regs) {
#pragma unroll
        for (int t = 0; t < JS; ++t) soc_one(tv + soc_o[t], soc_d[t]);
      } else {
        for (int cI = wvid; cI < D.nsoc; cI += BS / 64) soc_one(tv + D.soc_off[cI], D.soc_dim[cI]);
      }
      if constexpr (PSD) { batch_project_psd(D, tv, psd_ws, wvid, lane); batch_project_cone3<BS>(D, tv); }   // disjoint rows: no barrier needed between the cone kinds
      __syncthreads();
      // (side 17 .. 64 never reaches this kernel: build_lds_images gives such batches to the LDS-image kernel, whose registers have room for the
      //  block-Jacobi code)
#pragma unroll
      for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; if (i < m) sv[j] = tv[i]; }
      __syncthreads();
    }
  };
  // the accelerator (shared code above batch_admm_body): this thread's elements of w = [x ; rows] are its registers
  AaRegs S;
  AaMem M;
  if constexpr (AA) { M = aa_mem_of(D, k); aa_load(S, M.aa); }
  auto each = [&](auto&& fn) {
#pragma unroll
    for (int j = 0; j < JN; ++j) { const int i = OWN(j); if (i >= 0) fn(i, wx[j], wpx[j]); }
#pragma unroll
    for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; if (i < m) fn(n + i, wsv[j], wps[j]); }
  };

  if (do_init) solve_and_update();                                        // solver.jl:137-138
  long long it = ctl->iter;
  int status = 0;
  real o_cost = ctl->cost, o_rp = ctl->r_prim, o_rd = ctl->r_dual, o_mp = ctl->max_norm_prim, o_md = ctl->max_norm_dual;
  while (it < iter_target && it + S.sg < P.max_iter) {
    ++it;
    if constexpr (AA) {                                                   // acceleration_pre!; delta_y of the certificates (solver.jl:145-148)
      aa_pre<BS>(S, M, P, it, each, red);
      if (S.inf_due && !S.success) {
#pragma unroll
        for (int j = 0; j < JM; ++j) { const int i = tid + BS * j; if (i < m) D.inf_dy[om + i] = rhov[j] * (wps[j] - sv[j]); }
      }
    }
    admm_z();
    // ---- apply_rho_adaptation_rules! (solver.jl:242-282, parameters.jl:53-92); with an accelerator at the next non-accelerated iteration ----
    bool do_rho = P.adaptive_rho && P.adaptive_rho_interval > 0 && (it % P.adaptive_rho_interval) == 0 && (long long)(n_rho - 1) < P.max_adaptions;
    if constexpr (AA) {
      if (do_rho) S.rho_due = 1;
      do_rho = S.rho_due && !S.success;
      if (do_rho) S.rho_due = 0;
    }
    if (do_rho) {
      residuals(false);
      const real rpn = rp / (mp + R(1e-10)), rdn = rd / (md + R(1e-10));
      real nr = rho_s * sqrt(rpn / (rdn + R(1e-10)));
      nr = clamp_keep_nan(nr, P.rho_min, P.rho_max);
      const bool adapt = (nr > P.adapt_tol * rho_s) || (nr < (R(1.0) / P.adapt_tol) * rho_s);
      if (adapt) {
#pragma unroll
        for (int j = 0; j < JM; ++j) {
          const int i = tid + BS * j;
          if (i < m) {
            const int cc = cls[i]; real r2 = nr;
            if (cc == 1) r2 = r2 * P.rho_eq; else if (cc == 2) r2 = P.rho_min;
            rhov[j] = r2;
            wsv[j] = (R(1.0) / r2) * muv[j] + sv[j];
          }
        }
        if (SORTED) {
#pragma unroll
          for (int j = 0; j < JM; ++j) {
            if (ra[j] >= 0) { const int cc = cls[ra[j]]; real r2 = nr; if (cc == 1) r2 = r2 * P.rho_eq; else if (cc == 2) r2 = P.rho_min; rhoc[j] = r2; }
          }
        }
        if (tid == 0 && n_rho < COSMO_HIP_MAX_RHO_UPDATES) ctl->rho_updates[n_rho] = nr;
        rho_s = nr; n_rho += 1;
        if constexpr (AA) { S.iter = 0; S.init = 1; }                     // CA.restart! (solver.jl:272-275)
      }
    }
    solve_and_update();
    if constexpr (AA) {                                                   // acceleration_post! (accelerator_interface.jl:85-116): safeguarding
      if (S.active && S.success) {
        if (P.aa_safeguard) {
          if (aa_declined<BS>(S, M, P, each, red)) {
            aa_reset(M, each);
            admm_z();
            solve_and_update();
            S.sg += 1; S.n_decl += 1;
          } else S.n_ok += 1;
        }
        S.n_acc += 1;
      }
    }
    // ---- check_termination! (solver.jl:306-321) ----
    if ((it % P.check_termination) == 0 || it == 1) {
      residuals(P.unscale != 0);
      int st = 0;
      if (fabs(cost) > R(1e20)) st = COSMO_HIP_UNSOLVED;
      else if (rp < P.eps_abs + P.eps_rel * mp && rd < P.eps_abs + P.eps_rel * md &&
               ((P.obj_true != P.obj_true) || fabs(P.obj_true - cost) <= P.obj_true_tol)) st = COSMO_HIP_SOLVED;   // has_converged (residuals.jl:131-139)
      if constexpr (AA) {                                                 // check_activation!(ws, ::AccuracyActivation, r)
        if (st == 0 && !S.active && P.aa_start_acc >= R(0.0) &&
            rp < P.aa_start_acc + P.aa_start_acc * mp && rd < P.aa_start_acc + P.aa_start_acc * md) S.active = 1;
      }
      o_cost = cost; o_rp = rp; o_rd = rd; o_mp = mp; o_md = md; status = st;
      if (st != 0) break;
    }
    if constexpr (AA) {                                                   // solver.jl:326-349: the certificates wait for a non-accelerated iteration
      if (P.check_inf > 0 && (it % P.check_inf) == 0) S.inf_due = 1;
      else if (S.inf_due && !S.success) { S.inf_due = 0; S.need_inf = 1; break; }
    }
  }
  // iter (+ safeguarding_iter) == max_iter: calculate_result_info! and Max_iter_reached, overriding a status decided in that iteration
  // (solver.jl:173-176, reference quirk kept)
  if (it + S.sg >= P.max_iter) {
    residuals(P.unscale != 0);
    o_rp = rp; o_rd = rd; o_mp = mp; o_md = md; status = COSMO_HIP_MAX_ITER_REACHED;
  }
  if constexpr (AA) { if (tid == 0) aa_store(S, M.aa); }
#ifdef COSMO_BATCH_TIMING
  if (blockIdx.x == 0 && tid == 0) { g_bt[4] += clock64() - bt_kernel_t0; g_bt[5] += it; }
#endif
  // ---- store the persistent state; recover_mu! (solver.jl:167) ----
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int i = OWN(j);
    if (i >= 0) { D.w[onm + i] = wx[j]; D.w_prev[onm + i] = wpx[j]; D.x_tl[on + i] = xtl[j]; }
  }
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    const int i = tid + BS * j;
    if (i < m) {
      D.w[onm + n + i] = wsv[j]; D.w_prev[onm + n + i] = wps[j]; D.s[om + i] = sv[j]; D.rho[om + i] = rhov[j];
      D.mu[om + i] = rhov[j] * (wps[j] - sv[j]);
    }
  }
  if (tid == 0) {
    ctl->iter = it; ctl->solves = solves; ctl->kkt_iters_total = kkt_total; ctl->n_rho_updates = n_rho; ctl->rho = rho_s;
    ctl->cost = o_cost; ctl->r_prim = o_rp; ctl->r_dual = o_rd; ctl->max_norm_prim = o_mp; ctl->max_norm_dual = o_md; ctl->status = status;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Infeasibility certificates in batch mode (src/solver.jl:145-148, 326-349; src/infeasibility.jl:1-68).  The persistent kernels keep
// their registers for the iteration; the certificates run every check_infeasibility iterations only, so the host cuts the persistent
// launch at those iterations (cosmo_hip_batch_optimize):  ... iteration k ci | k_batch_inf_capture | iteration k ci + 1 |
// k_batch_inf_check | ...   Both kernels work on the state the persistent kernels leave in global memory (w, w_prev, s, rho), one
// workgroup per problem, all scalar tests inside the workgroup in the reference's order; a decided problem gets its status (and
// cost = +-Inf, solver.jl:339,345) and is skipped by every later launch.  Cones: ZeroSet / Nonnegatives / Box / SecondOrderCone: in_dual,
// in_pol_recc, support_function of src/convexset.jl:30-36, 76-82, 116-122, 850-861, 928-936; PSD cones of side <= 64: the smallest-eigenvalue
// tests of :415-424 (psd16.h / psdwg.h); exponential / power cones and duals: in_dual of the negated vector (:603-605, 724-726, 772; cone3.h).
// Accelerated batches run the check kernel only on the problems whose workgroup asked for it (BAa::need_inf).
// ---------------------------------------------------------------------------------------------------------------------
// delta_y at the top of the iteration that follows a flagged one: dy = mu = rho .* (w_prev_s - s)            (solver.jl:145-148)
__device__ __forceinline__ void batch_inf_capture_body(const BatchDev& D, const int k) {
  if (D.ctl[k].status != 0) return;
  const int n = D.n, m = D.m;
  const long long om = (long long)k * m, onm = (long long)k * (n + m);
  for (int i = threadIdx.x; i < m; i += COSMO_BS) D.inf_dy[om + i] = D.rho[om + i] * (D.w_prev[onm + n + i] - D.s[om + i]);
}
__global__ __launch_bounds__(COSMO_BS) void k_batch_inf_capture(BatchDev D) { batch_inf_capture_body(D, (int)blockIdx.x); }
// MERGED launches over several one-problem batches of different structure (batch_multi_optimize below): workgroup c takes the descriptor Ds[c] of ITS
// batch -- dimensions, cone tables, matrix and iterate pointers -- from global memory instead of the kernel arguments, and runs problem 0 of it
__global__ __launch_bounds__(COSMO_BS) void k_batch_inf_capture_multi(const BatchDev* __restrict__ Ds) { batch_inf_capture_body(Ds[blockIdx.x], 0); }

// flagged_only (accelerated batches): only the problems whose workgroup left its launch for this test (BAa::need_inf), see batch_admm_body
__device__ __forceinline__ void batch_inf_check_body(const BatchDev& D, const int k, real epi, real edi, int flagged_only) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  __shared__ int flag;
  __shared__ __attribute__((aligned(16))) unsigned char psd_ws[(COSMO_BS / 64) * PSD16_WS_STRIDE];
  __shared__ int flag2;
  BCtl* ctl = D.ctl + k;
  if (ctl->status != 0) return;
  if (flagged_only) {
    const int need = D.aa[k].need_inf;
    __syncthreads();
    if (!need) return;
    if (threadIdx.x == 0) D.aa[k].need_inf = 0;
  }
  constexpr int BS = COSMO_BS;
  // PSD cones (side <= 16): is_pos_def!(sign * mat(v) + tol I) <=> lambda_min(sign * mat(v)) > -tol  (convexset.jl:415-424, algebra.jl:226-238),
  // the smallest eigenvalue from the same wave-level Jacobi as the projection (mode 1: v is only read)
  auto psd_violates = [&](real* v, real sign, real tol) -> int {
    int bad = 0;
    const int wv_ = threadIdx.x >> 6, lane_ = threadIdx.x & 63;
    const Psd16Ws ws = psd16_ws_at(psd_ws + (size_t)wv_ * PSD16_WS_STRIDE);
    for (int cI = wv_; cI < D.npsd; cI += BS / 64) {
      real lm = 0.0;
      (void)psd16_wave(v + D.psd_off[cI], D.psd_d[cI], D.psd_kind[cI], ws, lane_, 1, sign, &lm, nullptr);
      if (!(lm > -tol)) bad = 1;
      wave_lds_fence();
    }
    for (int cI = 0; cI < D.nmid; ++cI) {              // side 17 .. 64: the whole workgroup (uniform control flow: every thread is here)
      __syncthreads();
      real* g = D.psdG + (long long)k * D.psdG_stride + D.mid_goff[cI];
      const int d = D.mid_d[cI], kind = D.mid_kind[cI], ld = D.mid_ld[cI], ncp = D.mid_ncp[cI];
      const real c = psdwg_populate<BS>(v + D.mid_off[cI], d, kind, ld, ncp, sign, 1, g, red);
      (void)psdwg_jacobi<4>(g, ld, ncp / 8, d, c, R(0.125), 0, psd_ws, &flag2);
      __syncthreads();
      const real lm = psdwg_finish<BS>(v + D.mid_off[cI], d, kind, ld, ncp, c, g, 1, red);
      if (!(lm > -tol)) bad = 1;
    }
    return bad;
  };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = D.n, m = D.m;
  const long long on = (long long)k * n, om = (long long)k * m, onm = (long long)k * (n + m);
  const real *w = D.w + onm, *w_prev = D.w_prev + onm, *s = D.s + om, *rho = D.rho + om, *q = D.q + on, *b = D.b + om;
  const real *Dinv = D.Dinv + on, *Einv = D.Einv + om;
  const real *bl = D.box_l + (long long)k * D.nbox, *bu = D.box_u + (long long)k * D.nbox;
  real *dy = D.inf_dy + om, *dx = D.ls_x + on, *adx = D.tmp_m + om;      // ls_x / tmp_m: scratch between two launches (rebuilt by every solve)
  const real c = R(1.0) / D.cinv[k];
  // dy -= mu_new ; dx = w_x - w_prev_x ; ||E dy||_inf, ||D dx||_inf, <q, dx>                  (solver.jl:331-335, infeasibility.jl:5, 35, 39)
  real ndy = 0.0, ndx = 0.0, qdx = 0.0;
  for (int i = tid; i < n + m; i += BS) {
    if (i < n) {
      const real d = w[i] - w_prev[i];
      dx[i] = d;
      ndx = amax(ndx, (R(1.0) / Dinv[i]) * d);
      qdx += q[i] * d;
    } else {
      const int r = i - n;
      const real mu = rho[r] * (w_prev[i] - s[r]);
      const real d = dy[r] - mu;
      dy[r] = d;
      ndy = amax(ndy, (R(1.0) / Einv[r]) * d);
    }
  }
  const real norm_dy = bmax<BS>(ndy, red), norm_dx = bmax<BS>(ndx, red), q_dx = bsum<BS>(qdx, red);
  __syncthreads();
  // ||Dinv P dx||_inf and ||Dinv A' dy||_inf in one pass over [P | A']                              (infeasibility.jl:12-17, 44-49)
  real a_p = 0.0, a_a = 0.0;
  { const CsrView PT = bview(D.PT, k);
    for (int t = 0; t < PT.nb; ++t)
      csr_stream_tile(PT, dx, dy, t, lds, red, [&](int row, real px, real aty) { const real d = Dinv[row]; a_p = amax(a_p, px * d); a_a = amax(a_a, aty * d); }); }
  const real pdx_norm = bmax<BS>(a_p, red), ady_norm = bmax<BS>(a_a, red);
  __syncthreads();
  // ---- is_primal_infeasible! (infeasibility.jl:1-29) ----
  if (norm_dy > epi && ady_norm <= epi * norm_dy) {
    if (tid == 0) flag = 0;
    __syncthreads();
    const real fneg = -R(1.0) / norm_dy;
    real dtb = 0.0, box = 0.0;
    int viol = 0;
    for (int i = tid; i < m; i += BS) {
      const real y = dy[i] * fneg;                     // delta_y *= (-1 / norm_dy)                   (:19)
      dy[i] = y;
      dtb += y * b[i];
      const uint32_t mt = D.meta[i], kind = mt & 3u;
      if (kind == 3u) { const uint32_t j = mt >> 2; box += (fabs(y) > epi && y > R(0.0)) ? y * bu[j] : y * bl[j]; }   // Box support function (convexset.jl:850-856)
      else if (kind == 2u) { if (-y < -epi) viol = 1; }                                                                // in_dual(-y) of Nonnegatives (:76-78)
    }
    __syncthreads();
    for (int cI = wv; cI < D.nsoc; cI += BS / 64) {   // in_dual(-y) of SecondOrderCone: ||y[2:]|| <= tol + (-y[1])   (:116-118)
      const real* x = dy + D.soc_off[cI]; const int d = D.soc_dim[cI];
      if (d == 0) continue;
      real a = 0.0;
      for (int i = 1 + lane; i < d; i += 64) { const real t = x[i]; a += t * t; }
      const real nx = sqrt(wave_sum(a));
      if (!(nx <= epi + (-x[0]))) viol = 1;
    }
    if ((D.npsd > 0 || D.nmid > 0) && psd_violates(dy, -R(1.0), epi)) viol = 1;          // in_dual!(-dyn) of the PSD cones (:415-418)
    for (int cI = tid; cI < D.n3; cI += BS) {                                             // in_dual(-dyn) of the 3-dimensional cones (:603-605, 724-726, 772)
      const real* p3 = dy + D.c3_off[cI];
      if (!cone3::in_dual_kind(cone3::V3{-p3[0], -p3[1], -p3[2]}, D.c3_kind[cI], D.c3_alpha[cI], epi)) viol = 1;
    }
    if (viol) atomicOr(&flag, 1);
    const real dyt_b = bsum<BS>(dtb, red), box_sf = bsum<BS>(box, red);          // (their barriers also publish `flag`)
    __syncthreads();
    const real sF = (flag ? (real)INFINITY : box_sf) - dyt_b;                     // support_function!: 0 if -y in the dual cone else Inf (:928-936)
    if (sF <= epi) {
      if (tid == 0) { ctl->status = COSMO_HIP_PRIMAL_INFEASIBLE; ctl->cost = (real)INFINITY; }   // solver.jl:337-340
      return;
    }
    __syncthreads();
  }
  // ---- is_dual_infeasible! (infeasibility.jl:32-68) ----
  if (norm_dx > edi && q_dx / (norm_dx * c) < -edi && pdx_norm / (norm_dx * c) <= edi) {
    if (tid == 0) flag = 0;
    __syncthreads();
    const real inv = R(1.0) / norm_dx;
    { const CsrView A = bview(D.A, k);
      for (int t = 0; t < A.nb; ++t)
        csr_stream_tile(A, dx, dx, t, lds, red, [&](int row, real s1, real s2) { adx[row] = ((s1 + s2) * Einv[row]) * inv; }); }   // (:53-59)
    __syncthreads();
    int viol = 0;
    for (int i = tid; i < m; i += BS) {               // in_pol_recc (convexset.jl:34-36, 80-82, 859-861)
      const real x = adx[i];
      const uint32_t mt = D.meta[i], kind = mt & 3u;
      if (kind == 1u) { if (fabs(x) > edi) viol = 1; }
      else if (kind == 2u) { if (x > edi) viol = 1; }
      else if (kind == 3u) { const uint32_t j = mt >> 2; if ((bu[j] == (real)INFINITY && x > edi) || (bl[j] == -(real)INFINITY && x < -edi)) viol = 1; }
    }
    for (int cI = wv; cI < D.nsoc; cI += BS / 64) {   // in_pol_recc of SecondOrderCone: ||x[2:]|| <= tol - x[1]   (:120-122)
      const real* x = adx + D.soc_off[cI]; const int d = D.soc_dim[cI];
      if (d == 0) continue;
      real a = 0.0;
      for (int i = 1 + lane; i < d; i += 64) { const real t = x[i]; a += t * t; }
      const real nx = sqrt(wave_sum(a));
      if (!(nx <= edi - x[0])) viol = 1;
    }
    if ((D.npsd > 0 || D.nmid > 0) && psd_violates(adx, -R(1.0), edi)) viol = 1;         // in_pol_recc!: is_neg_def (:421-424)
    for (int cI = tid; cI < D.n3; cI += BS) {                                             // in_pol_recc(x) = in_dual(-x) of the 3-dimensional cones
      const real* p3 = adx + D.c3_off[cI];
      if (!cone3::in_dual_kind(cone3::V3{-p3[0], -p3[1], -p3[2]}, D.c3_kind[cI], D.c3_alpha[cI], edi)) viol = 1;
    }
    if (viol) atomicOr(&flag, 1);
    __syncthreads();
    if (!flag && tid == 0) { ctl->status = COSMO_HIP_DUAL_INFEASIBLE; ctl->cost = -(real)INFINITY; }   // solver.jl:343-346
  }
}

__global__ __launch_bounds__(COSMO_BS) void k_batch_inf_check(BatchDev D, real epi, real edi, int flagged_only) { batch_inf_check_body(D, (int)blockIdx.x, epi, edi, flagged_only); }
__global__ __launch_bounds__(COSMO_BS) void k_batch_inf_check_multi(const BatchDev* __restrict__ Ds, real epi, real edi) { batch_inf_check_body(Ds[blockIdx.x], 0, epi, edi, 0); }
// the control blocks of the merged batches, gathered into one array for ONE copy to the host per slice
__global__ void k_batch_pack_ctl_multi(const BatchDev* __restrict__ Ds, int count, BCtl* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < count) out[c] = Ds[c].ctl[0];
}

// what the host loop of an accelerated batch needs after every launch: 16 bytes per problem instead of the whole BCtl (600 B) + BAa (2.2 KB) arrays
struct BState { long long iter; int status; int need_inf; };
__global__ void k_batch_pack_state(BatchDev D, BState* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.nprob) return;
  BState st;
  st.iter = D.ctl[k].iter; st.status = D.ctl[k].status; st.need_inf = D.aa ? D.aa[k].need_inf : 0;
  out[k] = st;
}

// warm start (solver.jl:128-129) for all problems
__global__ __launch_bounds__(COSMO_BS) void k_batch_set_w(BatchDev D, const real* __restrict__ x0, const real* __restrict__ s0,
                                                          const real* __restrict__ mu0) {
  const long long N = (long long)D.nprob * (D.n + D.m);
  for (long long g = (long long)blockIdx.x * COSMO_BS + threadIdx.x; g < N; g += (long long)gridDim.x * COSMO_BS) {
    const long long k = g / (D.n + D.m); const int i = (int)(g % (D.n + D.m));
    if (i < D.n) D.w[g] = x0 ? x0[k * D.n + i] : 0.0;
    else {
      const long long r = k * D.m + (i - D.n);
      const real sv = s0 ? s0[r] : 0.0, mv = mu0 ? mu0[r] : 0.0;
      D.w[g] = (R(1.0) / D.rho[r]) * mv + sv;
      D.s[r] = sv;
    }
    D.w_prev[g] = D.w[g];
  }
}

// =====================================================================================================================
// host side: C ABI of the batch mode
// =====================================================================================================================
struct cosmo_hip_batch {
  int device = 0; hipStream_t stream = nullptr; std::string err;
  int nprob = 0; long long n = 0, m = 0;
  std::vector<HostCsr> hA, hAT, hPT;             // staged per problem until finalize
  std::vector<real> hq, hb;
  std::vector<char> have;
  ConeTable cones; std::vector<real> hbox_l, hbox_u; int nbox = 0;
  cosmo_hip_params prm;
  bool finalized = false, have_cones = false, have_iterates = false;
  bool long_rows = false;                     // a row or a column of A with >= LONG_ROW entries in some problem: the LONG instantiations of the register kernel (build_lds_images)
  bool ext_cones = false;                     // the batch has cones beyond Zero / Nonnegatives / Box / SecondOrderCone (set_params): PSD of side >= 2, exponential, power
  BatchDev D;
  std::vector<void*> allocs;
  std::vector<real> hDinv, hEinv, hcinv;
  std::vector<int32_t> cls_host;
  long long iters_done = 0;
  // LDS-resident variant (build_lds_images): one image per problem, dynamic LDS = image + gather vectors + reduction slots
  unsigned char* d_img = nullptr; long long img_stride = 0; int lds_bytes = 0; int lds_bs = 0;
  int reg_mode = 0;    // 0: LdsOps kernel, 1: register-resident <512,1,2>, 2: <512,2,4>
  bool force_ext = false;
  void* d_state = nullptr; void* h_state = nullptr;     // accelerated host loop: {iter, status, need_inf} per problem (k_batch_pack_state)
  bool aa_on = false; cosmo_hip_accel_params aa_prm;      // cosmo_hip_batch_set_accelerator
  std::vector<int> h_permA, h_permT;          // compute assignment of the register kernel (build_lds_images), uploaded by set_params
  std::vector<int> h_posN, h_posM;            // positions of the gathered LDS vectors (build_lds_images)
  std::vector<int> h_qposA;                   // per problem: position of row i of A in the image's (sorted) storage order
  // sliced image of the register kernel (build_lds_images): first entry | length << 16 of every thread's rows / column; diagonal of P when P is diagonal in all problems
  std::vector<uint32_t> h_slA, h_slT; std::vector<real> h_pdiag; std::vector<unsigned char> h_pdiag_has;
};

static int32_t bfail(cosmo_hip_batch* b, int32_t code, const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (b) b->err = buf;
  return code;
}
#define BHIP(b, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return bfail((b), COSMO_HIP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); } while (0)

template <class T>
static int32_t bup(cosmo_hip_batch* b, const T** dptr, const std::vector<T>& v) {
  T* p = nullptr;
  BHIP(b, hipMalloc((void**)&p, std::max<size_t>(1, v.size()) * sizeof(T)));
  b->allocs.push_back(p);
  if (!v.empty()) BHIP(b, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dptr = p;
  return COSMO_HIP_OK;
}
template <class T>
static int32_t balloc(cosmo_hip_batch* b, T** dptr, size_t count) {
  T* p = nullptr;
  BHIP(b, hipMalloc((void**)&p, std::max<size_t>(1, count) * sizeof(T)));
  BHIP(b, hipMemset(p, 0, std::max<size_t>(1, count) * sizeof(T)));
  b->allocs.push_back(p);
  *dptr = p;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_create(cosmo_hip_batch** out, int32_t device_id, int64_t nprob, int64_t n, int64_t m) {
  if (!out || nprob <= 0 || n < 0 || m < 0) return COSMO_HIP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return COSMO_HIP_ERR_HIP;
  if (device_id < 0 || device_id >= ndev) return COSMO_HIP_ERR_INVALID;
  cosmo_hip_batch* b = new cosmo_hip_batch();
  b->device = device_id; b->nprob = (int)nprob; b->n = n; b->m = m;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&b->stream) != hipSuccess) { delete b; return COSMO_HIP_ERR_HIP; }
  b->hA.resize(nprob); b->hAT.resize(nprob); b->hPT.resize(nprob);
  b->hq.assign((size_t)nprob * n, 0.0); b->hb.assign((size_t)nprob * m, 0.0);
  b->have.assign(nprob, 0);
  b->hDinv.assign((size_t)nprob * n, 1.0); b->hEinv.assign((size_t)nprob * m, 1.0); b->hcinv.assign(nprob, 1.0);
  cosmo_hip_default_params(&b->prm);
  memset(&b->D, 0, sizeof b->D);
  *out = b;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_destroy(cosmo_hip_batch* b) {
  if (!b) return COSMO_HIP_OK;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (void* p : b->allocs) (void)hipFree(p);
  if (b->h_state) (void)hipHostFree(b->h_state);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return COSMO_HIP_OK;
}

extern "C" const char* cosmo_hip_batch_last_error(const cosmo_hip_batch* b) { return b ? b->err.c_str() : "null batch"; }

// CSC (Julia layout) -> host CSR of the matrix and of its transpose (same code path as the single-problem handle)
static int32_t bcsc(cosmo_hip_batch* b, int64_t nr, int64_t nc, const int64_t* colptr, const int64_t* rowval, const real* nzval,
                    HostCsr& Mt, HostCsr& M) {
  if (colptr[0] != 1) return bfail(b, COSMO_HIP_ERR_INVALID, "colptr must be 1-based");
  const int64_t nnz = colptr[nc] - 1;
  Mt.nrows = (int)nc; Mt.ncols = (int)nr; Mt.rowptr.resize(nc + 1); Mt.col.resize(nnz); Mt.val.resize(nnz);
  for (int64_t j = 0; j <= nc; ++j) Mt.rowptr[j] = (int)(colptr[j] - 1);
  std::vector<int> cnt(nr + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t i = rowval[k] - 1;
    if (i < 0 || i >= nr) return bfail(b, COSMO_HIP_ERR_INVALID, "row index out of range");
    Mt.col[k] = (int)i; Mt.val[k] = nzval[k]; cnt[i + 1]++;
  }
  M.nrows = (int)nr; M.ncols = (int)nc; M.rowptr.assign(nr + 1, 0);
  for (int64_t i = 0; i < nr; ++i) M.rowptr[i + 1] = M.rowptr[i] + cnt[i + 1];
  M.col.resize(nnz); M.val.resize(nnz);
  std::vector<int> pos(M.rowptr.begin(), M.rowptr.end() - 1);
  for (int64_t j = 0; j < nc; ++j)
    for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) { const int i = (int)(rowval[k] - 1); const int p = pos[i]++; M.col[p] = (int)j; M.val[p] = nzval[k]; }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_set_problem(cosmo_hip_batch* b, int64_t k, const int64_t* P_colptr, const int64_t* P_rowval,
                                               const real* P_nzval, const int64_t* A_colptr, const int64_t* A_rowval,
                                               const real* A_nzval, const real* q, const real* bvec) {
  if (!b || k < 0 || k >= b->nprob || b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_problem: bad call");
  const int64_t n = b->n, m = b->m;
  HostCsr Pt, Pm;
  int32_t rc = bcsc(b, n, n, P_colptr, P_rowval, P_nzval, Pt, Pm); if (rc) return rc;
  rc = bcsc(b, m, n, A_colptr, A_rowval, A_nzval, b->hAT[k], b->hA[k]); if (rc) return rc;
  HostCsr& PT = b->hPT[k];
  PT.nrows = (int)n; PT.ncols = (int)(n + m); PT.rowptr.assign(n + 1, 0); PT.split.assign(n, 0);
  PT.col.clear(); PT.val.clear();
  for (int64_t j = 0; j < n; ++j) {
    PT.rowptr[j] = (int)PT.col.size();
    for (int t = Pm.rowptr[j]; t < Pm.rowptr[j + 1]; ++t) { PT.col.push_back(Pm.col[t]); PT.val.push_back(Pm.val[t]); }
    PT.split[j] = (int)PT.col.size();
    for (int t = b->hAT[k].rowptr[j]; t < b->hAT[k].rowptr[j + 1]; ++t) { PT.col.push_back(b->hAT[k].col[t] + (int)n); PT.val.push_back(b->hAT[k].val[t]); }
  }
  PT.rowptr[n] = (int)PT.col.size();
  std::copy(q, q + n, b->hq.begin() + (size_t)k * n);
  std::copy(bvec, bvec + m, b->hb.begin() + (size_t)k * m);
  b->have[k] = 1;
  return COSMO_HIP_OK;
}

// same cone structure for every problem; Box bounds are per problem: box_l/box_u have nprob * nbox entries
extern "C" int32_t cosmo_hip_batch_set_cones_ex(cosmo_hip_batch* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                                const real* box_l, const real* box_u, const real* cone_param);
extern "C" int32_t cosmo_hip_batch_set_cones(cosmo_hip_batch* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                             const real* box_l, const real* box_u) {
  return cosmo_hip_batch_set_cones_ex(b, ncones, type, dim, box_l, box_u, nullptr);
}
// cone_param: alpha of PowerCone / DualPowerCone (as cosmo_hip_set_cones_ex); NULL = none
extern "C" int32_t cosmo_hip_batch_set_cones_ex(cosmo_hip_batch* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                                const real* box_l, const real* box_u, const real* cone_param) {
  if (!b || b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_cones: bad call");
  ConeTable& C = b->cones; C = ConeTable();
  int64_t off = 0, nbox = 0;
  for (int64_t k = 0; k < ncones; ++k) {
    if (type[k] == COSMO_HIP_PSD_SQUARE || type[k] == COSMO_HIP_PSD_TRIANGLE) {
      // side d from the dimension (convexset.jl:372: d = (isqrt(1 + 8 dim) - 1) / 2 for the triangle, isqrt(dim) for the square)
      long long d = 0;
      if (type[k] == COSMO_HIP_PSD_SQUARE) { while ((d + 1) * (d + 1) <= dim[k]) ++d; if (d * d != dim[k]) return bfail(b, COSMO_HIP_ERR_INVALID, "PsdCone: dimension %lld is not a square", (long long)dim[k]); }
      else { while ((d + 1) * (d + 2) / 2 <= dim[k]) ++d; if (d * (d + 1) / 2 != dim[k]) return bfail(b, COSMO_HIP_ERR_INVALID, "PsdConeTriangle: dimension %lld is not triangular", (long long)dim[k]); }
      if (d > 64) return bfail(b, COSMO_HIP_ERR_UNSUPPORTED, "batch mode projects PSD cones of side <= 64 (side %lld: use one handle per problem)", d);
    } else if (type[k] >= COSMO_HIP_EXP && type[k] <= COSMO_HIP_DUAL_POW) {
      if (dim[k] != 3) return bfail(b, COSMO_HIP_ERR_INVALID, "exponential / power cones have dimension 3");
      if (type[k] == COSMO_HIP_POW || type[k] == COSMO_HIP_DUAL_POW) {
        const double a = cone_param ? (double)cone_param[k] : 0.0;
        if (!(a > 0.0 && a < 1.0)) return bfail(b, COSMO_HIP_ERR_INVALID, "PowerCone: 0 < alpha < 1 (cone_param of batch_set_cones_ex)");
      }
    } else if (type[k] < COSMO_HIP_ZERO || type[k] > COSMO_HIP_SOC) return bfail(b, COSMO_HIP_ERR_UNSUPPORTED, "cone type %d", (int)type[k]);
    C.type.push_back(type[k]); C.dim.push_back(dim[k]); C.off.push_back(off); C.param.push_back(cone_param ? cone_param[k] : R(0.0)); off += dim[k];
    if (type[k] == COSMO_HIP_BOX) nbox += dim[k];
  }
  if (off != b->m) return bfail(b, COSMO_HIP_ERR_INVALID, "cone dimensions do not sum to m");
  b->nbox = (int)nbox;
  if (nbox) { b->hbox_l.assign(box_l, box_l + (size_t)b->nprob * nbox); b->hbox_u.assign(box_u, box_u + (size_t)b->nprob * nbox); }
  b->have_cones = true;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_set_scaling(cosmo_hip_batch* b, int64_t k, const real* Dinv, const real* Einv, double cinv) {
  if (!b || k < 0 || k >= b->nprob || b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_scaling: bad call");
  if (Dinv) std::copy(Dinv, Dinv + b->n, b->hDinv.begin() + (size_t)k * b->n);
  if (Einv) std::copy(Einv, Einv + b->m, b->hEinv.begin() + (size_t)k * b->m);
  b->hcinv[k] = cinv;
  return COSMO_HIP_OK;
}

static void brow_blocks(const std::vector<int>& rowptr, int nrows, std::vector<int>& rb) {
  rb.clear(); rb.push_back(0);
  const int ROWS_MAX = 4 * COSMO_BS;
  int r = 0;
  while (r < nrows) {
    int r1 = r; long long cnt = 0;
    while (r1 < nrows) {
      const long long rn = (long long)rowptr[r1 + 1] - rowptr[r1];
      if (r1 > r && (cnt + rn > COSMO_NNZ_PER_BLOCK || r1 - r >= ROWS_MAX)) break;
      cnt += rn; ++r1;
      if (cnt > COSMO_NNZ_PER_BLOCK) break;
    }
    rb.push_back(r1); r = r1;
  }
}

static int32_t bmat_upload(cosmo_hip_batch* b, std::vector<HostCsr>& Ms, BMat& out, int nrows, int split_col, bool has_split) {
  std::vector<int> rowptr, col, split, rb, rb_off, nb;
  std::vector<real> val;
  std::vector<long long> nz_off;
  for (int k = 0; k < b->nprob; ++k) {
    HostCsr& M = Ms[k];
    nz_off.push_back((long long)col.size());
    rowptr.insert(rowptr.end(), M.rowptr.begin(), M.rowptr.end());
    col.insert(col.end(), M.col.begin(), M.col.end());
    val.insert(val.end(), M.val.begin(), M.val.end());
    if (has_split) split.insert(split.end(), M.split.begin(), M.split.end());
    std::vector<int> r; brow_blocks(M.rowptr, nrows, r);
    rb_off.push_back((int)rb.size()); nb.push_back((int)r.size() - 1);
    for (size_t t = 0; t + 1 < r.size(); ++t) { rb.push_back(r[t]); rb.push_back(r[t + 1]); rb.push_back(M.rowptr[r[t]]); rb.push_back(M.rowptr[r[t + 1]]); }
    M = HostCsr();
  }
  out.nrows = nrows; out.split_col = split_col;
  int32_t rc;
  if ((rc = bup(b, &out.rowptr, rowptr))) return rc;
  if ((rc = bup(b, &out.col, col))) return rc;
  if ((rc = bup(b, &out.val, val))) return rc;
  if (has_split) { if ((rc = bup(b, &out.split, split))) return rc; } else out.split = nullptr;
  if ((rc = bup(b, &out.rb, rb))) return rc;
  if ((rc = bup(b, &out.nz_off, nz_off))) return rc;
  if ((rc = bup(b, &out.rb_off, rb_off))) return rc;
  if ((rc = bup(b, &out.nb, nb))) return rc;
  return COSMO_HIP_OK;
}

static void brow_blocks(const std::vector<int>& rowptr, int nrows, std::vector<int>& rb);

// the kernel instantiation a batch runs (one place: launch_batch_admm launches it, cosmo_hip_batch_kernel_info reports its registers / scratch)
struct BKernel { const void* fn; int bs; bool image; };
static BKernel batch_kernel_of(const cosmo_hip_batch* b) {
  bool psd = b->ext_cones;                               // the instantiations with the cones beyond Zero / Nonnegatives / Box / SecondOrderCone
  if (b->force_ext) psd = true;                           // COSMO_HIP_BATCH_EXT=1 (lab switch: that code is a run-time no-op without such cones)
  const bool img = b->d_img != nullptr;
#if !REAL_IS_FLOAT
  if (img && b->reg_mode == 1 && b->D.sliced) {
    if (b->aa_on) return {(const void*)k_batch_admm_reg<512, 1, 2, false, true, true>, 512, true};
    return {psd ? (const void*)k_batch_admm_reg<512, 1, 2, true, false, true> : (const void*)k_batch_admm_reg<512, 1, 2, false, false, true>, 512, true};
  }
#endif
  if (img && b->reg_mode == 1 && b->long_rows) {          // rows / columns of >= LONG_ROW entries: the cooperative passes (never with the sliced image)
    if (b->aa_on) return {(const void*)k_batch_admm_reg<512, 1, 2, false, true, false, true>, 512, true};
    return {psd ? (const void*)k_batch_admm_reg<512, 1, 2, true, false, false, true> : (const void*)k_batch_admm_reg<512, 1, 2, false, false, false, true>, 512, true};
  }
  if (img && b->reg_mode == 2 && b->long_rows) {
    if (b->aa_on) return {(const void*)k_batch_admm_reg<512, 2, 4, false, true, false, true>, 512, true};
    return {psd ? (const void*)k_batch_admm_reg<512, 2, 4, true, false, false, true> : (const void*)k_batch_admm_reg<512, 2, 4, false, false, false, true>, 512, true};
  }
  if (b->aa_on) {                    // accelerated loop: register kernel (batches without PSD cones), else the LDS-image kernel (512 threads) or the streaming kernel with the PSD code (a run-time no-op without such cones)
    if (img && b->reg_mode == 1) return {(const void*)k_batch_admm_reg<512, 1, 2, false, true>, 512, true};
    if (img && b->reg_mode == 2) return {(const void*)k_batch_admm_reg<512, 2, 4, false, true>, 512, true};
    if (img) return {(const void*)k_batch_admm_lds<512, true, true>, 512, true};
    return {(const void*)k_batch_admm<true, true>, COSMO_BS, false};
  }
  if (img && b->reg_mode == 1) return {psd ? (const void*)k_batch_admm_reg<512, 1, 2, true, false> : (const void*)k_batch_admm_reg<512, 1, 2, false, false>, 512, true};
  if (img && b->reg_mode == 2) return {psd ? (const void*)k_batch_admm_reg<512, 2, 4, true, false> : (const void*)k_batch_admm_reg<512, 2, 4, false, false>, 512, true};
  if (img) {
    if (psd) return {(const void*)k_batch_admm_lds<512, true, false>, 512, true};      // (build_lds_images fixed 512 threads for batches with PSD / exp / pow cones)
    if (b->lds_bs == 256) return {(const void*)k_batch_admm_lds<256, false, false>, 256, true};
    if (b->lds_bs == 512) return {(const void*)k_batch_admm_lds<512, false, false>, 512, true};
    return {(const void*)k_batch_admm_lds<1024, false, false>, 1024, true};
  }
  return {psd ? (const void*)k_batch_admm<true, false> : (const void*)k_batch_admm<false, false>, COSMO_BS, false};
}

// Builds the per-problem LDS images of k_batch_admm_lds if every problem fits (u16 indices, image + work vectors within the
// CU's LDS); otherwise leaves b->d_img = nullptr and the streaming kernel is used.  COSMO_HIP_BATCH_LDS=0 disables it,
// COSMO_HIP_BATCH_REG=0 keeps the iterates in global memory (LdsOps kernel), COSMO_HIP_BATCH_BS selects that kernel's
// workgroup size (256, 512 or 1024; default 512).
static int32_t build_lds_images(cosmo_hip_batch* b) {
  b->d_img = nullptr; b->lds_bs = 0;
  const char* e = getenv("COSMO_HIP_BATCH_LDS");
  if (e && atoi(e) == 0) return COSMO_HIP_OK;
  const long long n = b->n, m = b->m;
  if (n > 65535 || m > 65535 || n + m == 0) return COSMO_HIP_OK;
  int npsd = 0, nmid = 0, n3 = 0;
  for (size_t c = 0; c < b->cones.type.size(); ++c) if (b->cones.type[c] >= COSMO_HIP_EXP && b->cones.type[c] <= COSMO_HIP_DUAL_POW) n3 += 1;
  for (size_t c = 0; c < b->cones.type.size(); ++c)
    if ((b->cones.type[c] == COSMO_HIP_PSD_SQUARE || b->cones.type[c] == COSMO_HIP_PSD_TRIANGLE) && b->cones.dim[c] > 1) {
      npsd += 1;
      if (b->cones.dim[c] > (b->cones.type[c] == COSMO_HIP_PSD_SQUARE ? 16 * 16 : 16 * 17 / 2)) nmid += 1;        // side 17 .. 64: workgroup-level Jacobi
    }
  int bs = 512;
  if (const char* eb = getenv("COSMO_HIP_BATCH_BS")) { const int v = atoi(eb); if (v == 256 || v == 512 || v == 1024) bs = v; }
  { const char* ex = getenv("COSMO_HIP_BATCH_EXT"); b->force_ext = ex && atoi(ex) != 0; }     // measure the extended-cone instantiations on batches without such cones
  if (npsd > 0 || n3 > 0 || b->aa_on || b->force_ext) bs = 512;  // the extended-cone / accelerated instantiations of the LDS-image kernel exist for 512 threads
  // register-resident iterates (k_batch_admm_reg, 512 threads) when the vectors fit 1-2 (n) / 2-4 (m) elements per thread
  b->reg_mode = 0;
  { const char* er = getenv("COSMO_HIP_BATCH_REG");
    if (!(er && atoi(er) == 0)) {
      if (n <= 512 && m <= 1024) b->reg_mode = 1; else if (n <= 1024 && m <= 2048) b->reg_mode = 2;
      if (b->aa_on && (npsd > 0 || n3 > 0)) b->reg_mode = 0;   // the accelerated register kernel is instantiated without the PSD / exp / pow code (registers): the LDS-image kernel takes such batches
      if (nmid > 0) b->reg_mode = 0;            // the block-Jacobi code on top of ~200 live registers would spill: the LDS-image kernel (187 VGPRs) takes such batches
      if (b->reg_mode) bs = 512;
    } }
  int max_lds = 0;
  if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, b->device) != hipSuccess) return COSMO_HIP_OK;
  auto up16 = [](long long x) { return (x + 15) / 16 * 16; };
  std::vector<std::vector<unsigned char>> imgs((size_t)b->nprob);
  long long stride = 0;
  // sliced image (register kernel <512, 1, 2>, double precision; row_sliced): built next to the row-major one, used if every problem's fits
  std::vector<std::vector<unsigned char>> imgs2((size_t)b->nprob);
  long long stride2 = 0;
  const long long SL_WS0 = (long long)sizeof(real) * (512 + 2 * 512 + 2 * (512 / 64));      // xv, tv, block-sum slots in front of the image
  bool want_sliced = !REAL_IS_FLOAT && b->reg_mode == 1 && !(getenv("COSMO_HIP_BATCH_SLICED") && atoi(getenv("COSMO_HIP_BATCH_SLICED")) == 0);
  bool p_all_diag = true;                                                                    // P diagonal (or empty rows) in every problem: it moves into registers
  for (int k = 0; k < b->nprob && p_all_diag; ++k) {
    const HostCsr& PT = b->hPT[k];
    for (long long j = 0; j < n && p_all_diag; ++j) { const int len = PT.split[j] - PT.rowptr[j]; if (len > 1 || (len == 1 && PT.col[PT.rowptr[j]] != (int)j)) p_all_diag = false; }
  }
  // rows / columns of A with >= LONG_ROW entries (a dense budget row, an epigraph variable): the register kernel <512, 1, 2> hands them to a whole wave in
  // its Krylov passes (LONG instantiations, row_long_or_pipe3); the sliced image would pad the 32 rows of such a row's slice to its length: not built then
  b->long_rows = false;
  if (b->reg_mode != 0 && !(getenv("COSMO_HIP_BATCH_LONG") && atoi(getenv("COSMO_HIP_BATCH_LONG")) == 0)) {
    for (int k = 0; k < b->nprob && !b->long_rows; ++k) {
      const HostCsr &A = b->hA[k], &AT = b->hAT[k];
      for (long long i = 0; i < m && !b->long_rows; ++i) if (A.rowptr[i + 1] - A.rowptr[i] >= LONG_ROW) b->long_rows = true;
      for (long long j = 0; j < n && !b->long_rows; ++j) if (AT.rowptr[j + 1] - AT.rowptr[j] >= LONG_ROW) b->long_rows = true;
    }
    if (b->long_rows) want_sliced = false;
  }
  b->h_slA.clear(); b->h_slT.clear(); b->h_pdiag.clear(); b->h_pdiag_has.clear();
  // LDS-image kernel, register-CG form of its reduced solve (batch_admm_body, RCG: the 512-thread instantiations with the extended cones): the compute
  // assignment sorted by length (D.permA / D.permT in that form's <2, 4>-slot layout) and, with it, the image STORED in that order (round 6)
  const bool rcg_form = b->reg_mode == 0 && bs == 512 && n <= 2 * 512 && m <= 4 * 512 && (npsd > 0 || n3 > 0 || b->aa_on || b->force_ext) &&
                        !(getenv("COSMO_HIP_BATCH_LDSCG") && atoi(getenv("COSMO_HIP_BATCH_LDSCG")) == 0);
  const bool rcg_sorted = rcg_form && !(getenv("COSMO_HIP_BATCH_LDSCG_SORTED") && atoi(getenv("COSMO_HIP_BATCH_LDSCG_SORTED")) == 0);
  const bool rcg_stored = rcg_sorted && !(getenv("COSMO_HIP_BATCH_STORE_SORTED") && atoi(getenv("COSMO_HIP_BATCH_STORE_SORTED")) == 0);
  for (int k = 0; k < b->nprob; ++k) {
    const HostCsr &A = b->hA[k], &AT = b->hAT[k], &PT = b->hPT[k];
    const long long nnzA = (long long)A.val.size();
    long long nnzP = 0;
    for (long long j = 0; j < n; ++j) nnzP += PT.split[j] - PT.rowptr[j];
    if (nnzA > 65535 || nnzP > 65535) return COSMO_HIP_OK;
    std::vector<int> rA, rAT, rPT;
    // register-CG form: rows of A by decreasing length, columns by decreasing length of [P | A'] (stable: equal lengths stay in index order)
    std::vector<int> gordA, gordT, prA, prT, prPT;
    if (rcg_sorted) {
      gordA.resize((size_t)m); gordT.resize((size_t)n);
      for (long long i = 0; i < m; ++i) gordA[(size_t)i] = (int)i;
      for (long long j = 0; j < n; ++j) gordT[(size_t)j] = (int)j;
      std::stable_sort(gordA.begin(), gordA.end(), [&](int x, int y) { return A.rowptr[x + 1] - A.rowptr[x] > A.rowptr[y + 1] - A.rowptr[y]; });
      auto clen = [&](int j) { return (PT.split[j] - PT.rowptr[j]) + (AT.rowptr[j + 1] - AT.rowptr[j]); };
      std::stable_sort(gordT.begin(), gordT.end(), [&](int x, int y) { return clen(x) > clen(y); });
      if (b->h_permA.empty()) { b->h_permA.assign((size_t)b->nprob * 4 * 512, -1); b->h_permT.assign((size_t)b->nprob * 2 * 512, -1); }
      for (long long q = 0; q < m; ++q) {                      // position q: slot q / 512 of thread q % 512, odd slots backwards (as in the register kernel)
        const long long sl = q / 512, t = (sl & 1) ? 511 - q % 512 : q % 512;
        b->h_permA[(size_t)k * 4 * 512 + (size_t)(sl * 512 + t)] = gordA[(size_t)q];
      }
      for (long long q = 0; q < n; ++q) b->h_permT[(size_t)k * 2 * 512 + (size_t)q] = gordT[(size_t)q];
    }
    if (rcg_stored) {                                         // row pointers of the stored order: the tile lists of the generic loops are cut on them
      prA.assign((size_t)m + 1, 0); prT.assign((size_t)n + 1, 0); prPT.assign((size_t)n + 1, 0);
      for (long long q = 0; q < m; ++q) prA[(size_t)q + 1] = prA[(size_t)q] + (A.rowptr[gordA[(size_t)q] + 1] - A.rowptr[gordA[(size_t)q]]);
      for (long long q = 0; q < n; ++q) {
        const int c = gordT[(size_t)q];
        prT[(size_t)q + 1] = prT[(size_t)q] + (AT.rowptr[c + 1] - AT.rowptr[c]);
        prPT[(size_t)q + 1] = prPT[(size_t)q] + (PT.rowptr[c + 1] - PT.rowptr[c]);
      }
      brow_blocks(prA, (int)m, rA); brow_blocks(prT, (int)n, rAT); brow_blocks(prPT, (int)n, rPT);
    } else {
    brow_blocks(A.rowptr, (int)m, rA); brow_blocks(AT.rowptr, (int)n, rAT); brow_blocks(PT.rowptr, (int)n, rPT);
    }
    LdsHdr h; memset(&h, 0, sizeof h);
    h.nnzA = (int)nnzA; h.nnzP = (int)nnzP; h.nbA = (int)rA.size() - 1; h.nbAT = (int)rAT.size() - 1; h.nbPT = (int)rPT.size() - 1;
    long long o = up16(sizeof(LdsHdr));
    h.oAval = (int)o; o = up16(o + (long long)sizeof(real) * nnzA);
    h.oPval = (int)o; o = up16(o + (long long)sizeof(real) * nnzP);
    h.oRbA = (int)o; o = up16(o + 16LL * h.nbA);
    h.oRbAT = (int)o; o = up16(o + 16LL * h.nbAT);
    h.oRbPT = (int)o; o = up16(o + 16LL * h.nbPT);
    h.oArp = (int)o; o = up16(o + 2 * (m + 1));
    h.oAcol = (int)o; o = up16(o + 2 * nnzA);
    h.oTrp = (int)o; o = up16(o + 2 * (n + 1));
    h.oTpr = (int)o; o = up16(o + 4 * nnzA);
    h.oPrp = (int)o; o = up16(o + 2 * (n + 1));
    h.oPcol = (int)o; o = up16(o + 2 * nnzP);
    if (rcg_stored) { h.oAord = (int)o; o = up16(o + 2 * m); h.oTord = (int)o; o = up16(o + 2 * n); }
    h.bytes = (int)o;
    if (o + (long long)sizeof(real) * (n + m) + (long long)sizeof(real) * 2 * (bs / 64) > max_lds) return COSMO_HIP_OK;
    std::vector<unsigned char>& im = imgs[(size_t)k];
    im.assign((size_t)o, 0);
    memcpy(im.data(), &h, sizeof h);
    real* Aval = reinterpret_cast<real*>(im.data() + h.oAval);
    real* Pval = reinterpret_cast<real*>(im.data() + h.oPval);
    unsigned short* Arp = reinterpret_cast<unsigned short*>(im.data() + h.oArp);
    unsigned short* Acol = reinterpret_cast<unsigned short*>(im.data() + h.oAcol);
    unsigned short* Trp = reinterpret_cast<unsigned short*>(im.data() + h.oTrp);
    uint32_t* Tpr = reinterpret_cast<uint32_t*>(im.data() + h.oTpr);
    unsigned short* Prp = reinterpret_cast<unsigned short*>(im.data() + h.oPrp);
    unsigned short* Pcol = reinterpret_cast<unsigned short*>(im.data() + h.oPcol);
    for (long long t = 0; t < nnzA; ++t) { Aval[t] = A.val[t]; Acol[t] = (unsigned short)A.col[t]; }
    for (long long i = 0; i <= m; ++i) Arp[i] = (unsigned short)A.rowptr[i];
    std::vector<int> cur(A.rowptr.begin(), A.rowptr.end() - 1);
    long long pp = 0;
    for (long long j = 0; j < n; ++j) {
      Trp[j] = (unsigned short)AT.rowptr[j]; Prp[j] = (unsigned short)pp;
      for (int t = PT.rowptr[j]; t < PT.split[j]; ++t) { Pval[pp] = PT.val[t]; Pcol[pp] = (unsigned short)PT.col[t]; ++pp; }
      for (int t = AT.rowptr[j]; t < AT.rowptr[j + 1]; ++t) {
        const int i = AT.col[t]; const int p = cur[i]++;
        if (p >= A.rowptr[i + 1] || A.col[p] != (int)j || A.val[p] != AT.val[t]) return bfail(b, COSMO_HIP_ERR_INVALID, "LDS image: A / A' mismatch");
        Tpr[t] = (uint32_t)p | ((uint32_t)i << 16);
      }
    }
    Trp[n] = (unsigned short)AT.rowptr[n]; Prp[n] = (unsigned short)pp;
    auto fill_rb = [&](int off, const std::vector<int>& r, const std::vector<int>& rowptr) {
      int* d = reinterpret_cast<int*>(im.data() + off);
      for (size_t t = 0; t + 1 < r.size(); ++t) { d[4 * t] = r[t]; d[4 * t + 1] = r[t + 1]; d[4 * t + 2] = rowptr[r[t]]; d[4 * t + 3] = rowptr[r[t + 1]]; }
    };
    if (rcg_stored) {
      // the arrays filled above in index order move to the stored order: same values, same order inside every row -- every row sum keeps its bits
      std::vector<real> nAval((size_t)nnzA), nPval((size_t)nnzP);
      std::vector<unsigned short> nAcol((size_t)nnzA), nPcol((size_t)nnzP);
      std::vector<uint32_t> nTpr((size_t)nnzA);
      std::vector<int> newstart((size_t)m);
      long long wq = 0;
      for (long long q = 0; q < m; ++q) {
        const int r = gordA[(size_t)q];
        newstart[(size_t)r] = (int)wq;
        for (int t = A.rowptr[r]; t < A.rowptr[r + 1]; ++t) { nAval[(size_t)wq] = Aval[t]; nAcol[(size_t)wq] = Acol[t]; ++wq; }
      }
      long long wt = 0, wp = 0;
      for (long long q = 0; q < n; ++q) {
        const int c = gordT[(size_t)q];
        for (int t = Trp[c]; t < Trp[c + 1]; ++t) {
          const uint32_t pr = Tpr[t]; const int row = (int)(pr >> 16), pold = (int)(pr & 0xffffu);
          nTpr[(size_t)wt++] = (uint32_t)(newstart[(size_t)row] + (pold - A.rowptr[row])) | ((uint32_t)row << 16);
        }
        for (int t = Prp[c]; t < Prp[c + 1]; ++t) { nPval[(size_t)wp] = Pval[t]; nPcol[(size_t)wp] = Pcol[t]; ++wp; }
      }
      std::copy(nAval.begin(), nAval.end(), Aval); std::copy(nAcol.begin(), nAcol.end(), Acol);
      std::copy(nTpr.begin(), nTpr.end(), Tpr); std::copy(nPval.begin(), nPval.end(), Pval); std::copy(nPcol.begin(), nPcol.end(), Pcol);
      for (long long q = 0; q <= m; ++q) Arp[q] = (unsigned short)prA[(size_t)q];
      long long pq = 0;
      for (long long q = 0; q < n; ++q) { Trp[q] = (unsigned short)prT[(size_t)q]; Prp[q] = (unsigned short)pq; pq += PT.split[gordT[(size_t)q]] - PT.rowptr[gordT[(size_t)q]]; }
      Trp[n] = (unsigned short)prT[(size_t)n]; Prp[n] = (unsigned short)pq;
      unsigned short* Aord = reinterpret_cast<unsigned short*>(im.data() + h.oAord);
      unsigned short* Tord = reinterpret_cast<unsigned short*>(im.data() + h.oTord);
      for (long long q = 0; q < m; ++q) Aord[q] = (unsigned short)gordA[(size_t)q];
      for (long long q = 0; q < n; ++q) Tord[q] = (unsigned short)gordT[(size_t)q];
      fill_rb(h.oRbA, rA, prA); fill_rb(h.oRbAT, rAT, prT); fill_rb(h.oRbPT, rPT, prPT);
    } else {
    fill_rb(h.oRbA, rA, A.rowptr); fill_rb(h.oRbAT, rAT, AT.rowptr); fill_rb(h.oRbPT, rPT, PT.rowptr);
    }
    stride = std::max(stride, o);
    const char* e_sorted = getenv("COSMO_HIP_BATCH_SORTED");      // =0: every thread computes the rows it owns (the form until round 3)
    if (b->reg_mode == 1 && !(e_sorted && atoi(e_sorted) == 0)) {
      // length-sorted compute assignment (see k_batch_admm_reg): position p of the sorted order goes to slot p / 512 of thread p % 512,
      // i.e. a wave-step covers 64 consecutive positions
      const int JMs = b->reg_mode == 1 ? 2 : 4, JNs = b->reg_mode == 1 ? 1 : 2;
      if (b->h_permA.empty()) { b->h_permA.assign((size_t)b->nprob * JMs * 512, -1); b->h_permT.assign((size_t)b->nprob * JNs * 512, -1); }
      std::vector<int> ord((size_t)m);
      for (long long i = 0; i < m; ++i) ord[(size_t)i] = (int)i;
      std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return A.rowptr[x + 1] - A.rowptr[x] > A.rowptr[y + 1] - A.rowptr[y]; });
      // rows: slots alternate direction (position q of slot s goes to thread q % 512 for even s, 511 - q % 512 for odd s), so that the wave
      // with the longest rows of one slot has the shortest of the next: the pass ends with its slowest wave
      for (long long q = 0; q < m; ++q) {
        const long long sl = q / 512, t = (sl & 1) ? 511 - q % 512 : q % 512;
        b->h_permA[(size_t)k * JMs * 512 + (size_t)(sl * 512 + t)] = ord[(size_t)q];
      }
      const std::vector<int> ordA = ord;
      ord.resize((size_t)n);
      for (long long j = 0; j < n; ++j) ord[(size_t)j] = (int)j;
      auto clen = [&](int j) { return (PT.split[j] - PT.rowptr[j]) + (AT.rowptr[j + 1] - AT.rowptr[j]); };
      std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return clen(x) > clen(y); });
      for (long long q = 0; q < n; ++q) b->h_permT[(size_t)k * JNs * 512 + (size_t)q] = ord[(size_t)q];
      // STORAGE ORDER of the image = the sorted order (round 5): the rows of A, the columns of A' and the rows of P are laid out in the order of the
      // compute assignment, so that the 64 rows a wave works on in one trip are NEIGHBOURS in the value / index arrays -- lane l reads at (start of
      // its row) + t with starts about one row length apart: a constant-stride access, which the LDS serves at 4.1 (b64) / 2.2 (u16) cycles per
      // wave-read against 7.2 for the scattered reads that CSR order gives a length-sorted assignment (bench/lds_conflict_lab.hip).  Arp / Trp / Prp
      // are indexed by sorted POSITION from here on (the register kernel derives the positions of its rows / columns from the thread index, and
      // reads the position of the rows it owns from D.qposA); same values, same per-row order: every row sum keeps its bits.
      if (!(getenv("COSMO_HIP_BATCH_STORE_SORTED") && atoi(getenv("COSMO_HIP_BATCH_STORE_SORTED")) == 0)) {
        if (b->h_qposA.empty()) b->h_qposA.assign((size_t)b->nprob * m, 0);
        std::vector<real> nAval((size_t)nnzA); std::vector<unsigned short> nAcol((size_t)nnzA), nArp((size_t)m + 1);
        std::vector<int> newstart((size_t)m);
        long long w = 0;
        for (long long q = 0; q < m; ++q) {
          const int r = ordA[(size_t)q];
          nArp[(size_t)q] = (unsigned short)w; newstart[(size_t)r] = (int)w;
          b->h_qposA[(size_t)k * m + (size_t)r] = (int)q;
          for (int t = A.rowptr[r]; t < A.rowptr[r + 1]; ++t) { nAval[(size_t)w] = Aval[t]; nAcol[(size_t)w] = Acol[t]; ++w; }
        }
        nArp[(size_t)m] = (unsigned short)w;
        std::vector<uint32_t> nTpr((size_t)nnzA); std::vector<unsigned short> nTrp((size_t)n + 1), nPrp((size_t)n + 1), nPcol((size_t)nnzP);
        std::vector<real> nPval((size_t)nnzP);
        long long wt = 0, wp = 0;
        for (long long q = 0; q < n; ++q) {
          const int c = ord[(size_t)q];
          nTrp[(size_t)q] = (unsigned short)wt; nPrp[(size_t)q] = (unsigned short)wp;
          for (int t = Trp[c]; t < Trp[c + 1]; ++t) {
            const uint32_t pr = Tpr[t]; const int row = (int)(pr >> 16), pold = (int)(pr & 0xffffu);
            nTpr[(size_t)wt++] = (uint32_t)(newstart[(size_t)row] + (pold - A.rowptr[row])) | ((uint32_t)row << 16);
          }
          for (int t = Prp[c]; t < Prp[c + 1]; ++t) { nPval[(size_t)wp] = Pval[t]; nPcol[(size_t)wp] = Pcol[t]; ++wp; }
        }
        nTrp[(size_t)n] = (unsigned short)wt; nPrp[(size_t)n] = (unsigned short)wp;
        std::copy(nAval.begin(), nAval.end(), Aval); std::copy(nAcol.begin(), nAcol.end(), Acol); std::copy(nArp.begin(), nArp.end(), Arp);
        std::copy(nTpr.begin(), nTpr.end(), Tpr); std::copy(nTrp.begin(), nTrp.end(), Trp); std::copy(nPrp.begin(), nPrp.end(), Prp);
        std::copy(nPval.begin(), nPval.end(), Pval); std::copy(nPcol.begin(), nPcol.end(), Pcol);
      }
      // Positions of the gathered vectors.  A ds_read_b64 is served 32 lanes at a time and an 8-byte slot p lies in bank pair p mod 32:
      // the 32 rows a half-wave works on in one step gather 32 entries, and every extra entry on a busy bank pair costs an LDS cycle
      // (random indices: ~3.4 per group).  Greedy assignment: entries by decreasing number of appearances, each to the bank pair that
      // adds the fewest conflicts over the groups it appears in (capacity = slots of that residue), slots handed out in order.
      const char* e_color = getenv("COSMO_HIP_BATCH_COLOR");
      if (!(e_color && atoi(e_color) == 0)) {
        if (b->h_posN.empty()) { b->h_posN.assign((size_t)b->nprob * n, 0); b->h_posM.assign((size_t)b->nprob * m, 0); }
        auto assign_positions = [&](int nent, const std::vector<std::vector<int>>& groups, int* pos) {
          std::vector<std::vector<int>> where((size_t)nent);                 // entry -> groups it appears in
          for (size_t g = 0; g < groups.size(); ++g) for (int e : groups[g]) where[(size_t)e].push_back((int)g);
          std::vector<int> order((size_t)nent);
          for (int e = 0; e < nent; ++e) order[(size_t)e] = e;
          std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return where[(size_t)x].size() > where[(size_t)y].size(); });
          std::vector<int> cap(32, 0), used(32, 0);
          for (int q = 0; q < nent; ++q) cap[q % 32] += 1;
          std::vector<unsigned char> cnt(groups.size() * 32, 0);
          for (int e : order) {
            int best = -1; long long bestc = 0;
            for (int bk = 0; bk < 32; ++bk) {
              if (used[bk] >= cap[bk]) continue;
              long long c = 0;
              for (int g : where[(size_t)e]) c += cnt[(size_t)g * 32 + bk];
              if (best < 0 || c < bestc || (c == bestc && used[bk] < used[best])) { best = bk; bestc = c; }
            }
            pos[e] = best + 32 * used[best];
            used[best] += 1;
            for (int g : where[(size_t)e]) cnt[(size_t)g * 32 + best] += 1;
          }
        };
        // groups of the x-gathers: rows computed by a half-wave in one step of the A pass; columns of P gathered by a half-wave in the column pass
        std::vector<std::vector<int>> gN, gM;
        for (int sl = 0; sl < JMs; ++sl)
          for (int hw = 0; hw < 16; ++hw) {
            int maxlen = 0;
            for (int t = 32 * hw; t < 32 * hw + 32; ++t) { const int r = b->h_permA[(size_t)k * JMs * 512 + (size_t)sl * 512 + t]; if (r >= 0) maxlen = std::max(maxlen, A.rowptr[r + 1] - A.rowptr[r]); }
            for (int st = 0; st < maxlen; ++st) {
              std::vector<int> g;
              for (int t = 32 * hw; t < 32 * hw + 32; ++t) { const int r = b->h_permA[(size_t)k * JMs * 512 + (size_t)sl * 512 + t]; if (r >= 0 && A.rowptr[r] + st < A.rowptr[r + 1]) g.push_back(A.col[A.rowptr[r] + st]); }
              if (g.size() > 1) gN.push_back(g);
            }
          }
        for (int sl = 0; sl < JNs; ++sl)
          for (int hw = 0; hw < 16; ++hw) {
            int maxp = 0, maxt = 0;
            for (int t = 32 * hw; t < 32 * hw + 32; ++t) {
              const int c = b->h_permT[(size_t)k * JNs * 512 + (size_t)sl * 512 + t];
              if (c >= 0) { maxp = std::max(maxp, PT.split[c] - PT.rowptr[c]); maxt = std::max(maxt, AT.rowptr[c + 1] - AT.rowptr[c]); }
            }
            for (int st = 0; st < maxp; ++st) {
              std::vector<int> g;
              for (int t = 32 * hw; t < 32 * hw + 32; ++t) { const int c = b->h_permT[(size_t)k * JNs * 512 + (size_t)sl * 512 + t]; if (c >= 0 && PT.rowptr[c] + st < PT.split[c]) g.push_back(PT.col[PT.rowptr[c] + st]); }
              if (g.size() > 1) gN.push_back(g);
            }
            for (int st = 0; st < maxt; ++st) {
              std::vector<int> g;
              for (int t = 32 * hw; t < 32 * hw + 32; ++t) { const int c = b->h_permT[(size_t)k * JNs * 512 + (size_t)sl * 512 + t]; if (c >= 0 && AT.rowptr[c] + st < AT.rowptr[c + 1]) g.push_back(AT.col[AT.rowptr[c] + st]); }
              if (g.size() > 1) gM.push_back(g);
            }
          }
        int* pN = b->h_posN.data() + (size_t)k * n;
        int* pM = b->h_posM.data() + (size_t)k * m;
        assign_positions((int)n, gN, pN);
        assign_positions((int)m, gM, pM);
        // the image's gather indices become positions
        std::vector<unsigned char>& im2 = imgs[(size_t)k];
        unsigned short* Acol2 = reinterpret_cast<unsigned short*>(im2.data() + h.oAcol);
        uint32_t* Tpr2 = reinterpret_cast<uint32_t*>(im2.data() + h.oTpr);
        unsigned short* Pcol2 = reinterpret_cast<unsigned short*>(im2.data() + h.oPcol);
        for (long long t = 0; t < nnzA; ++t) { Acol2[t] = (unsigned short)pN[Acol2[t]]; Tpr2[t] = (Tpr2[t] & 0xffffu) | ((uint32_t)pM[Tpr2[t] >> 16] << 16); }
        for (long long t = 0; t < nnzP; ++t) Pcol2[t] = (unsigned short)pN[Pcol2[t]];
      }
      // ---- the sliced image of this problem (see row_sliced): from the final arrays above (stored in sorted order, gather indices = positions) ----
      if (want_sliced && b->h_qposA.empty()) want_sliced = false;                 // needs the sorted storage order (Arp / Trp / Prp indexed by position)
      if (want_sliced) {
        if (b->h_slA.empty()) {
          b->h_slA.assign((size_t)b->nprob * 2 * 512, 0u); b->h_slT.assign((size_t)b->nprob * 512, 0u);
          if (p_all_diag) { b->h_pdiag.assign((size_t)b->nprob * n, R(0.0)); b->h_pdiag_has.assign((size_t)b->nprob * n, 0); }
        }
        auto qA = [&](int j, int t) { return 512 * j + ((j & 1) ? 511 - t : t); };        // slot j of thread t computes the row at this sorted position
        long long offA[32], LA[32], offT[16], LT[16], curA = 0, curT = 0, needA = 0, needT = 0;
        for (int j = 0; j < 2; ++j) for (int sx = 0; sx < 16; ++sx) {
          long long L = 0;
          for (int l = 0; l < 32; ++l) { const int q = qA(j, 32 * sx + l); if (q < m) L = std::max<long long>(L, Arp[q + 1] - Arp[q]); }
          offA[j * 16 + sx] = curA; LA[j * 16 + sx] = L; curA += 32 * L;
        }
        for (int sx = 0; sx < 16; ++sx) {
          long long L = 0;
          for (int l = 0; l < 32; ++l) { const int q = 32 * sx + l; if (q < n) L = std::max<long long>(L, Trp[q + 1] - Trp[q]); }
          offT[sx] = curT; LT[sx] = L; curT += 32 * L;
        }
        // a lane runs to its WAVE's longest row and reads two trips ahead: into the slices behind its own, or into the zero tail sized here
        for (int j = 0; j < 2; ++j) for (int sx = 0; sx < 16; ++sx) needA = std::max(needA, offA[j * 16 + sx] + 32 * (std::max(LA[j * 16 + (sx & ~1)], LA[j * 16 + (sx | 1)]) + 2));
        for (int sx = 0; sx < 16; ++sx) needT = std::max(needT, offT[sx] + 32 * (std::max(LT[sx & ~1], LT[sx | 1]) + 2));
        const long long nS = std::max(curA, needA), nT = std::max(curT, needT);
        LdsHdr h2; memset(&h2, 0, sizeof h2);
        h2.nnzA = (int)nS; h2.nnzP = p_all_diag ? 0 : (int)nnzP;
        long long o2 = up16(sizeof(LdsHdr));
        h2.oAval = (int)o2; o2 = up16(o2 + (long long)sizeof(real) * nS);
        h2.oTpr = (int)o2; o2 = up16(o2 + 4 * nT);
        h2.oAcol = (int)o2; o2 = up16(o2 + 2 * nS);
        if (!p_all_diag) {
          h2.oPval = (int)o2; o2 = up16(o2 + (long long)sizeof(real) * nnzP);
          h2.oPrp = (int)o2; o2 = up16(o2 + 2 * (n + 1));
          h2.oPcol = (int)o2; o2 = up16(o2 + 2 * nnzP);
        }
        h2.bytes = (int)o2;
        if (nS > 65535 || nT > 65535 || SL_WS0 + o2 > max_lds || SL_WS0 + h2.oAval + (long long)sizeof(real) * nS >= (1LL << 18)) want_sliced = false;
        else {
          std::vector<unsigned char>& i2 = imgs2[(size_t)k];
          i2.assign((size_t)o2, 0);
          memcpy(i2.data(), &h2, sizeof h2);
          real* Aval2 = reinterpret_cast<real*>(i2.data() + h2.oAval);
          unsigned short* Acol2s = reinterpret_cast<unsigned short*>(i2.data() + h2.oAcol);
          uint32_t* Tpr2s = reinterpret_cast<uint32_t*>(i2.data() + h2.oTpr);
          std::vector<long long> newpos((size_t)std::max<long long>(nnzA, 1), 0);
          for (int j = 0; j < 2; ++j) for (int sx = 0; sx < 16; ++sx) for (int l = 0; l < 32; ++l) {
            const int t = 32 * sx + l, q = qA(j, t);
            const long long off = offA[j * 16 + sx] + l;
            const long long start = q < m ? Arp[q] : 0, len = q < m ? Arp[q + 1] - Arp[q] : 0;
            b->h_slA[((size_t)k * 2 + (size_t)j) * 512 + (size_t)t] = (uint32_t)off | ((uint32_t)len << 16);
            for (long long tt = 0; tt < len; ++tt) {
              const long long e = off + 32 * tt;
              Aval2[e] = Aval[start + tt]; Acol2s[e] = (unsigned short)(Acol[start + tt] * (unsigned)sizeof(real)); newpos[(size_t)(start + tt)] = e;
            }
          }
          for (int sx = 0; sx < 16; ++sx) for (int l = 0; l < 32; ++l) {
            const int q = 32 * sx + l;
            const long long off = offT[sx] + l;
            const long long start = q < n ? Trp[q] : 0, len = q < n ? Trp[q + 1] - Trp[q] : 0;
            b->h_slT[(size_t)k * 512 + (size_t)q] = (uint32_t)off | ((uint32_t)len << 16);
            for (long long tt = 0; tt < len; ++tt) {
              const uint32_t pr = Tpr[start + tt];
              Tpr2s[off + 32 * tt] = (uint32_t)(SL_WS0 + h2.oAval + (long long)sizeof(real) * newpos[(size_t)(pr & 0xffffu)]) | ((pr >> 16) << 21);
            }
          }
          if (!p_all_diag) {
            memcpy(i2.data() + h2.oPval, Pval, (size_t)nnzP * sizeof(real));
            memcpy(i2.data() + h2.oPrp, Prp, (size_t)(n + 1) * 2);
            memcpy(i2.data() + h2.oPcol, Pcol, (size_t)nnzP * 2);
          } else {
            for (long long j = 0; j < n; ++j) if (PT.split[j] > PT.rowptr[j]) { b->h_pdiag[(size_t)k * n + (size_t)j] = PT.val[PT.rowptr[j]]; b->h_pdiag_has[(size_t)k * n + (size_t)j] = 1; }
          }
          stride2 = std::max(stride2, o2);
        }
      }
    }
  }
  // the sliced image replaces the row-major one if every problem's fits (with at least one wave workspace of the small PSD cones where the batch has such
  // cones) and the sliced instantiation has no static LDS in front of the dynamic block (its gathered vectors must start at LDS address 0)
  bool use_sliced = want_sliced && !b->h_slA.empty() && b->reg_mode == 1;
#if REAL_IS_FLOAT
  use_sliced = false;
#else
  if (use_sliced && npsd > 0 && (max_lds - up16(SL_WS0 + stride2)) / PSD16_WS_STRIDE < 1) use_sliced = false;
  if (use_sliced) {
    const void* fs = b->aa_on ? (const void*)k_batch_admm_reg<512, 1, 2, false, true, true>
                              : ((npsd > 0 || n3 > 0 || b->force_ext) ? (const void*)k_batch_admm_reg<512, 1, 2, true, false, true> : (const void*)k_batch_admm_reg<512, 1, 2, false, false, true>);
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fs) != hipSuccess || fa.sharedSizeBytes != 0) { (void)hipGetLastError(); use_sliced = false; }
  }
#endif
  if (use_sliced) { imgs.swap(imgs2); stride = stride2; }
  else { b->h_slA.clear(); b->h_slT.clear(); b->h_pdiag.clear(); b->h_pdiag_has.clear(); }
  b->D.sliced = use_sliced ? 1 : 0;
  unsigned char* d = nullptr;
  BHIP(b, hipMalloc((void**)&d, (size_t)stride * b->nprob));
  b->allocs.push_back(d);
  BHIP(b, hipMemset(d, 0, (size_t)stride * b->nprob));
  for (int k = 0; k < b->nprob; ++k)
    BHIP(b, hipMemcpy(d + (size_t)k * stride, imgs[(size_t)k].data(), imgs[(size_t)k].size(), hipMemcpyHostToDevice));
  // wave workspaces of the small PSD cones behind the reduction slots: as many as fit, at most one per wave; none fits => streaming kernel
  const long long ws_base = use_sliced ? up16(SL_WS0 + stride) : ((stride + (long long)sizeof(real) * (n + m + 2 * (bs / 64)) + 15) / 16) * 16;
  int nws = 0;
  if (npsd > 0) {
    nws = (int)std::min<long long>(bs / 64, (max_lds - ws_base) / PSD16_WS_STRIDE);
    if (nws < (nmid > 0 ? 4 : 1)) { b->reg_mode = 0; return COSMO_HIP_OK; }       // (the image is released with the batch); side 17 .. 64 needs four (one per block pair)
  }
  b->D.psd_nws = nws;
  { const char* ec = getenv("COSMO_HIP_BATCH_LDSCG");
    b->D.regcg = (bs == 512 && n <= 2 * 512 && m <= 4 * 512 && !(ec && atoi(ec) == 0)) ? 1 : 0; }
  if (!b->D.regcg || b->reg_mode != 0) { if (rcg_sorted) { b->h_permA.clear(); b->h_permT.clear(); } }
  b->d_img = d; b->img_stride = stride; b->lds_bs = bs;
  b->lds_bytes = (int)(npsd > 0 ? ws_base + (long long)nws * PSD16_WS_STRIDE
                                : (use_sliced ? SL_WS0 + stride : stride + (long long)sizeof(real) * (n + m) + (long long)sizeof(real) * 2 * (bs / 64)));
  const void* fn = batch_kernel_of(b).fn;                  // (d_img, reg_mode, sliced, lds_bs are set: the instantiation launch_batch_admm will launch)
  // The attribute belongs to the kernel INSTANTIATION, which is process-global, not to this batch: a batch group builds several classes that share an
  // instantiation with different lds_bytes, from several threads.  It is therefore set to the most the instantiation can be granted -- the device maximum
  // minus the instantiation's own static LDS (every lds_bytes is <= max_lds by construction) -- so that a later, smaller class cannot lower it under an
  // earlier one's launch on a runtime that enforces the value.  (Round 6, first form: the plain device maximum -- refused for the
  // instantiations with a few bytes of static LDS, and every batch with PSD / exponential / power cones fell back to the streaming kernel; found with
  // cosmo_hip_batch_kernel_info, pinned by tests/test_gpu_batch.py::test_extended_cone_batches_run_the_lds_image_kernel.)
  { hipFuncAttributes fa;
    int grant = 0;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess) grant = max_lds - (int)fa.sharedSizeBytes;
    if (grant < b->lds_bytes || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, grant) != hipSuccess) {
      (void)hipGetLastError();
      b->d_img = nullptr; b->reg_mode = 0;          // the device does not grant that much LDS: streaming kernel
    } }
  return COSMO_HIP_OK;
}

static int32_t launch_batch_admm(cosmo_hip_batch* b, const BParams& P, long long target, int do_init) {
  const BKernel K = batch_kernel_of(b);
  BatchDev D = b->D; BParams Pc = P; long long tg = target; int di = do_init;
  const unsigned char* img = b->d_img; long long stride = b->img_stride;
  void* args[6] = {&D, &Pc, &tg, &di, &img, &stride};         // (the streaming kernels take the first four)
  BHIP(b, hipLaunchKernel(K.fn, dim3(b->nprob), dim3(K.bs), args, K.image ? (size_t)b->lds_bytes : 0, b->stream));
  BHIP(b, hipGetLastError());
  return COSMO_HIP_OK;
}

// set_params finalises the batch: uploads everything, classifies rows, builds the rho vectors (set_rho_vec!)
extern "C" int32_t cosmo_hip_batch_set_params(cosmo_hip_batch* b, const cosmo_hip_params* p) {
  if (!b || !p) return COSMO_HIP_ERR_INVALID;
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (!b->have_cones) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_cones first");
  for (int k = 0; k < b->nprob; ++k) if (!b->have[k]) return bfail(b, COSMO_HIP_ERR_INVALID, "problem %d not set", k);
  if (p->kkt_kind != COSMO_HIP_KKT_CG) return bfail(b, COSMO_HIP_ERR_UNSUPPORTED, "batch mode implements the CG KKT solver");
  if (p->adaptive_rho && p->adaptive_rho_interval == 0) return bfail(b, COSMO_HIP_ERR_UNSUPPORTED, "adaptive_rho_interval == 0");
  if (b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch already finalised");
  b->prm = *p;
  const int nprob = b->nprob; const long long n = b->n, m = b->m;
  BatchDev& D = b->D;
  D.nprob = nprob; D.n = (int)n; D.m = (int)m;
  int32_t rc;
  D.permA = nullptr; D.permT = nullptr; D.posN = nullptr; D.posM = nullptr; D.qposA = nullptr;
  D.slA = nullptr; D.slT = nullptr; D.pdiag = nullptr; D.pdiag_has = nullptr; D.sliced = 0;
  // (until round 6 the launch looked at the cones of side <= 16 and the three-dimensional cones only: a batch whose ONLY such cones were PSD cones of side
  //  17 .. 64 would have run an instantiation without the PSD code)
  b->ext_cones = false;
  for (size_t c = 0; c < b->cones.type.size(); ++c) {
    const int ty = b->cones.type[c];
    if (ty >= COSMO_HIP_EXP && ty <= COSMO_HIP_DUAL_POW) b->ext_cones = true;
    if ((ty == COSMO_HIP_PSD_SQUARE || ty == COSMO_HIP_PSD_TRIANGLE) && b->cones.dim[c] > 1) b->ext_cones = true;
  }
  if ((rc = build_lds_images(b))) return rc;               // needs the host CSR copies that bmat_upload releases
  if (b->d_img && (b->reg_mode == 1 || (b->reg_mode == 0 && D.regcg)) && !b->h_permA.empty()) {        // (no image: the streaming kernel runs and needs none of this)
    if ((rc = bup(b, &D.permA, b->h_permA))) return rc;
    if ((rc = bup(b, &D.permT, b->h_permT))) return rc;
    if (!b->h_qposA.empty()) { if ((rc = bup(b, &D.qposA, b->h_qposA))) return rc; }
    if (D.sliced) {
      if ((rc = bup(b, &D.slA, b->h_slA))) return rc;
      if ((rc = bup(b, &D.slT, b->h_slT))) return rc;
      if (!b->h_pdiag.empty()) { if ((rc = bup(b, &D.pdiag, b->h_pdiag))) return rc; if ((rc = bup(b, &D.pdiag_has, b->h_pdiag_has))) return rc; }
    }
    if (!b->h_posN.empty()) {
      if ((rc = bup(b, &D.posN, b->h_posN))) return rc;
      if ((rc = bup(b, &D.posM, b->h_posM))) return rc;
    }
  }
  if ((rc = bmat_upload(b, b->hA, D.A, (int)m, (int)n, false))) return rc;
  if ((rc = bmat_upload(b, b->hAT, D.AT, (int)n, (int)m, false))) return rc;
  if ((rc = bmat_upload(b, b->hPT, D.PT, (int)n, (int)n, true))) return rc;
  if ((rc = bup(b, &D.q, b->hq))) return rc;
  if ((rc = bup(b, &D.b, b->hb))) return rc;
  if ((rc = bup(b, &D.Dinv, b->hDinv))) return rc;
  if ((rc = bup(b, &D.Einv, b->hEinv))) return rc;
  if ((rc = bup(b, &D.cinv, b->hcinv))) return rc;
  // cone metadata (shared) + per-problem classification
  const ConeTable& C = b->cones;
  std::vector<uint32_t> meta((size_t)m, 0u);
  std::vector<int> soc_off, soc_dim, psd_off, psd_d, psd_kind, mid_off, mid_d, mid_kind, mid_ld, mid_ncp;
  std::vector<long long> mid_goff;
  std::vector<int> c3_off, c3_kind;
  std::vector<real> c3_alpha;
  long long gtot = 0;
  long long boxp = 0;
  for (size_t k = 0; k < C.type.size(); ++k) {
    const long long o = C.off[k], d = C.dim[k];
    switch (C.type[k]) {
      case COSMO_HIP_ZERO: for (long long i = 0; i < d; ++i) meta[o + i] = 1u; break;
      case COSMO_HIP_NONNEG: for (long long i = 0; i < d; ++i) meta[o + i] = 2u; break;
      case COSMO_HIP_PSD_SQUARE: case COSMO_HIP_PSD_TRIANGLE:
        if (d == 1) { meta[o] = 2u; break; }                        // the 1-D case is max(x, 0) (convexset.jl:303-305, 404-405)
        { long long sd = 0;
          if (C.type[k] == COSMO_HIP_PSD_SQUARE) { while ((sd + 1) * (sd + 1) <= d) ++sd; } else { while ((sd + 1) * (sd + 2) / 2 <= d) ++sd; }
          if (sd <= 16) { psd_off.push_back((int)o); psd_d.push_back((int)sd); psd_kind.push_back((int)C.type[k]); }
          else {                                                    // geometry as psd.hip: psd_plan_create
            int ld = (((int)sd + 15) / 16) * 16, nb = ((int)sd + 7) / 8;
            if (nb & 1) nb += 1;
            int ncp = nb * 8;
            if (ncp < ld) ncp = ld;
            mid_off.push_back((int)o); mid_d.push_back((int)sd); mid_kind.push_back((int)C.type[k]); mid_ld.push_back(ld); mid_ncp.push_back(ncp);
            mid_goff.push_back(gtot); gtot += (long long)ld * ncp;
          } }
        break;
      case COSMO_HIP_BOX: for (long long i = 0; i < d; ++i) meta[o + i] = 3u | ((uint32_t)(boxp + i) << 2); boxp += d; break;
      case COSMO_HIP_SOC: soc_off.push_back((int)o); soc_dim.push_back((int)d); break;
      case COSMO_HIP_EXP: case COSMO_HIP_DUAL_EXP: case COSMO_HIP_POW: case COSMO_HIP_DUAL_POW:
        c3_off.push_back((int)o); c3_kind.push_back((int)C.type[k]); c3_alpha.push_back(k < C.param.size() ? C.param[k] : R(0.0)); break;
    }
  }
  D.n3 = (int)c3_off.size();
  if ((rc = bup(b, &D.c3_off, c3_off)) || (rc = bup(b, &D.c3_kind, c3_kind)) || (rc = bup(b, &D.c3_alpha, c3_alpha))) return rc;
  b->cls_host.assign((size_t)nprob * m, 0);
  const real big = p->cosmo_infty_min_scaling;
  for (int k = 0; k < nprob; ++k) {
    long long bp = 0;
    for (size_t c = 0; c < C.type.size(); ++c) {
      const long long o = C.off[c], d = C.dim[c];
      int32_t* cl = b->cls_host.data() + (size_t)k * m + o;
      if (C.type[c] == COSMO_HIP_ZERO) for (long long i = 0; i < d; ++i) cl[i] = 1;
      else if (C.type[c] == COSMO_HIP_NONNEG) { for (long long i = 0; i < d; ++i) if (b->hb[(size_t)k * m + o + i] > big) cl[i] = 2; }
      else if (C.type[c] == COSMO_HIP_BOX) {
        for (long long i = 0; i < d; ++i) {
          const real l = b->hbox_l[(size_t)k * b->nbox + bp + i], u = b->hbox_u[(size_t)k * b->nbox + bp + i];
          cl[i] = (l < -big && u > big) ? 2 : (((u - l) < (real)p->rho_tol) ? 1 : 0);
        }
        bp += d;
      }
    }
  }
  std::vector<real> rho0((size_t)nprob * m);
  for (size_t i = 0; i < rho0.size(); ++i) {
    real rv = p->rho;
    if (b->cls_host[i] == 1) rv = rv * (real)p->rho_eq_over_rho_ineq; else if (b->cls_host[i] == 2) rv = (real)p->rho_min;
    rho0[i] = rv;
  }
  if ((rc = bup(b, &D.meta, meta))) return rc;
  if ((rc = bup(b, &D.box_l, b->hbox_l))) return rc;
  if ((rc = bup(b, &D.box_u, b->hbox_u))) return rc;
  D.nbox = b->nbox;
  D.nsoc = (int)soc_off.size();
  if ((rc = bup(b, &D.soc_off, soc_off))) return rc;
  if ((rc = bup(b, &D.soc_dim, soc_dim))) return rc;
  D.npsd = (int)psd_off.size();
  if ((rc = bup(b, &D.psd_off, psd_off)) || (rc = bup(b, &D.psd_d, psd_d)) || (rc = bup(b, &D.psd_kind, psd_kind))) return rc;
  if (!b->d_img) D.psd_nws = COSMO_BS / 64;                    // streaming kernel: static workspaces for all of its four waves
  D.nmid = (int)mid_off.size();
  if ((rc = bup(b, &D.mid_off, mid_off)) || (rc = bup(b, &D.mid_d, mid_d)) || (rc = bup(b, &D.mid_kind, mid_kind)) || (rc = bup(b, &D.mid_ld, mid_ld)) ||
      (rc = bup(b, &D.mid_ncp, mid_ncp)) || (rc = bup(b, &D.mid_goff, mid_goff))) return rc;
  D.psdG_stride = gtot; D.psdG = nullptr;
  if (gtot > 0 && (rc = balloc(b, &D.psdG, (size_t)gtot * nprob))) return rc;
  std::vector<int> cls32(b->cls_host.begin(), b->cls_host.end());
  if ((rc = bup(b, &D.rho_cls, cls32))) return rc;
  const size_t NM = (size_t)nprob * (n + m), Nn = (size_t)nprob * n, Nm = (size_t)nprob * m;
  if ((rc = balloc(b, &D.w, NM)) || (rc = balloc(b, &D.w_prev, NM)) || (rc = balloc(b, &D.s, Nm)) || (rc = balloc(b, &D.mu, Nm)) ||
      (rc = balloc(b, &D.s_tl, Nm)) || (rc = balloc(b, &D.ls_s, Nm)) || (rc = balloc(b, &D.y2, Nm)) || (rc = balloc(b, &D.tmp_m, Nm)) ||
      (rc = balloc(b, &D.nu, Nm)) || (rc = balloc(b, &D.rho, Nm)) || (rc = balloc(b, &D.ls_x, Nn)) || (rc = balloc(b, &D.x_tl, Nn)) ||
      (rc = balloc(b, &D.rhs, Nn)) || (rc = balloc(b, &D.r, Nn)) || (rc = balloc(b, &D.u, Nn)) || (rc = balloc(b, &D.c, Nn)) ||
      (rc = balloc(b, &D.inf_dy, Nm))) return rc;
  D.aa_mem = 0; D.aa = nullptr; D.aa_G = D.aa_Q = D.aa_f = D.aa_fl = D.aa_gl = nullptr;
  if (b->aa_on) {                                           // mem = min(mem, dim), as the single-problem accelerator
    D.aa_mem = (int)std::min<long long>(b->aa_prm.mem, std::max<long long>(n + m, 1));
    if ((rc = balloc(b, &D.aa, (size_t)nprob)) || (rc = balloc(b, &D.aa_G, NM * D.aa_mem)) || (rc = balloc(b, &D.aa_Q, NM * D.aa_mem)) ||
        (rc = balloc(b, &D.aa_f, NM)) || (rc = balloc(b, &D.aa_fl, NM)) || (rc = balloc(b, &D.aa_gl, NM))) return rc;
  }
  BHIP(b, hipMemcpy(D.rho, rho0.data(), Nm * sizeof(real), hipMemcpyHostToDevice));
  std::vector<BCtl> ctl0(nprob);
  memset(ctl0.data(), 0, sizeof(BCtl) * nprob);
  for (auto& c : ctl0) { c.rho = p->rho; c.n_rho_updates = 1; c.rho_updates[0] = p->rho; c.cost = INFINITY; c.r_prim = INFINITY; c.r_dual = INFINITY; }
  if ((rc = balloc(b, &D.ctl, (size_t)nprob))) return rc;
  BHIP(b, hipMemcpy(D.ctl, ctl0.data(), sizeof(BCtl) * nprob, hipMemcpyHostToDevice));
  // tolerance schedule tol_constant / k^tol_exponent (get_tolerance, kktsolver_indirect.jl:168-170), host libm like the large path
  const long long tl = std::min<long long>(std::max<long long>(p->max_iter + 2, 16), 4000000);
  std::vector<real> tt((size_t)tl);
  for (long long k = 0; k < tl; ++k) tt[k] = p->tol_constant / pow((real)(k + 1), p->tol_exponent);
  if ((rc = bup(b, &D.tol_table, tt))) return rc;
  D.tol_len = tl;
  b->finalized = true;
  b->hq.clear(); b->hq.shrink_to_fit();
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_get_rho_classes(cosmo_hip_batch* b, int64_t k, int32_t* cls) {
  if (!b || !b->finalized || k < 0 || k >= b->nprob || !cls) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_get_rho_classes: bad call");
  std::copy(b->cls_host.begin() + (size_t)k * b->m, b->cls_host.begin() + (size_t)(k + 1) * b->m, cls);
  return COSMO_HIP_OK;
}

// x0, s0, mu0: nprob*n, nprob*m, nprob*m (NULL = zeros)
extern "C" int32_t cosmo_hip_batch_set_iterates(cosmo_hip_batch* b, const real* x0, const real* s0, const real* mu0) {
  if (!b || !b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_iterates: set_params first");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  const size_t Nn = (size_t)b->nprob * b->n, Nm = (size_t)b->nprob * b->m;
  real *dx = nullptr, *ds = nullptr, *dm = nullptr;
  if (x0) { BHIP(b, hipMalloc((void**)&dx, Nn * sizeof(real))); BHIP(b, hipMemcpy(dx, x0, Nn * sizeof(real), hipMemcpyHostToDevice)); }
  if (s0) { BHIP(b, hipMalloc((void**)&ds, Nm * sizeof(real))); BHIP(b, hipMemcpy(ds, s0, Nm * sizeof(real), hipMemcpyHostToDevice)); }
  if (mu0) { BHIP(b, hipMalloc((void**)&dm, Nm * sizeof(real))); BHIP(b, hipMemcpy(dm, mu0, Nm * sizeof(real), hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(k_batch_set_w, dim3(2048), dim3(COSMO_BS), 0, b->stream, b->D, dx, ds, dm);
  BHIP(b, hipStreamSynchronize(b->stream));
  if (dx) (void)hipFree(dx); if (ds) (void)hipFree(ds); if (dm) (void)hipFree(dm);
  // optimize! restarts at iter = 0 with status undetermined; KKT counters persist
  std::vector<BCtl> c(b->nprob);
  BHIP(b, hipMemcpy(c.data(), b->D.ctl, sizeof(BCtl) * b->nprob, hipMemcpyDeviceToHost));
  for (auto& x : c) { x.status = 0; x.iter = 0; x.cost = INFINITY; x.r_prim = INFINITY; x.r_dual = INFINITY; x.max_norm_prim = 0; x.max_norm_dual = 0; }
  BHIP(b, hipMemcpy(b->D.ctl, c.data(), sizeof(BCtl) * b->nprob, hipMemcpyHostToDevice));
  if (b->D.aa) {                                            // optimize! restarts the accelerator (src/setup.jl:47-49): empty memory, inactive, counters zero
    std::vector<BAa> a0((size_t)b->nprob);
    memset(a0.data(), 0, sizeof(BAa) * a0.size());
    for (auto& a : a0) a.init_phase = 1;
    BHIP(b, hipMemcpy(b->D.aa, a0.data(), sizeof(BAa) * a0.size(), hipMemcpyHostToDevice));
  }
  b->have_iterates = true; b->iters_done = 0;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_set_accelerator(cosmo_hip_batch* b, const cosmo_hip_accel_params* p) {
  if (!b) return COSMO_HIP_ERR_INVALID;
  if (b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_set_accelerator: call before batch_set_params (the kernel variant is chosen there)");
  b->aa_on = false;
  if (!p || p->kind == COSMO_HIP_ACCEL_EMPTY) return COSMO_HIP_OK;
  if (p->kind != COSMO_HIP_ACCEL_ANDERSON) return bfail(b, COSMO_HIP_ERR_UNSUPPORTED, "accelerator kind %d", (int)p->kind);
  if (p->mem < 1 || p->mem > BAA_MEM || p->min_mem < 1 || p->start_iter < 2 || !(p->safeguard_tol >= 0.0) || !(p->eta_max > 0.0))
    return bfail(b, p->mem > BAA_MEM ? COSMO_HIP_ERR_UNSUPPORTED : COSMO_HIP_ERR_INVALID,
                 "batch_set_accelerator: need 1 <= mem <= %d (batch mode), min_mem >= 1, start_iter >= 2", BAA_MEM);
  b->aa_prm = *p; b->aa_on = true;
  return COSMO_HIP_OK;
}

// per problem {accelerated steps, safeguard accepted, safeguard declined, memory restarts, active, safeguarding_iter}
extern "C" int32_t cosmo_hip_batch_get_accel_stats(cosmo_hip_batch* b, int64_t* out) {
  if (!b || !b->finalized || !out) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_get_accel_stats: bad call");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  for (long long i = 0; i < 6LL * b->nprob; ++i) out[i] = 0;
  if (!b->D.aa) return COSMO_HIP_OK;
  std::vector<BAa> a((size_t)b->nprob);
  BHIP(b, hipMemcpyAsync(a.data(), b->D.aa, sizeof(BAa) * a.size(), hipMemcpyDeviceToHost, b->stream));
  BHIP(b, hipStreamSynchronize(b->stream));
  for (int k = 0; k < b->nprob; ++k) {
    out[6 * k] = a[k].accelerated; out[6 * k + 1] = a[k].accepted; out[6 * k + 2] = a[k].declined; out[6 * k + 3] = a[k].restarts;
    out[6 * k + 4] = a[k].active; out[6 * k + 5] = a[k].sg_iter;
  }
  return COSMO_HIP_OK;
}

static BParams bparams(const cosmo_hip_batch* b, bool certificates) {
  const cosmo_hip_params& p = b->prm;
  BParams P;
  memset(&P, 0, sizeof P);
  P.aa_start_acc = -1.0;
  if (b->aa_on) {
    const cosmo_hip_accel_params& a = b->aa_prm;
    P.aa_start_iter = a.start_iter; P.aa_min_mem = a.min_mem; P.aa_safeguard = a.safeguard; P.aa_tau = (real)a.safeguard_tol; P.aa_eta_max = (real)a.eta_max;
    P.aa_start_acc = (real)a.start_accuracy;
    const long long ci = p.check_infeasibility;
    P.check_inf = (certificates && ci > 0 && ci < (1LL << 40)) ? ci : 0;
  }
  P.sigma = p.sigma; P.alpha = p.alpha; P.eps_abs = p.eps_abs; P.eps_rel = p.eps_rel; P.obj_true = p.obj_true; P.obj_true_tol = p.obj_true_tol; P.rho_min = p.rho_min; P.rho_max = p.rho_max;
  P.rho_eq = p.rho_eq_over_rho_ineq; P.adapt_tol = p.adaptive_rho_tolerance; P.max_iter = p.max_iter;
  P.max_adaptions = p.adaptive_rho_max_adaptions; P.check_termination = p.check_termination; P.adaptive_rho = p.adaptive_rho;
  P.adaptive_rho_interval = p.adaptive_rho_interval; P.unscale = p.unscale_residuals;
  return P;
}

// Runs every problem to a status (or max_iter).  results: nprob entries.  The loop is launched in slices of
// `check_termination`-aligned iterations so that the host can enforce time_limit.
extern "C" int32_t cosmo_hip_batch_optimize(cosmo_hip_batch* b, cosmo_hip_result* results) {
  if (!b || !b->have_iterates || !results) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_optimize: set_iterates first");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  const BParams P = bparams(b, true);
  const auto t0 = std::chrono::steady_clock::now();
  const long long slice = std::max<long long>(b->prm.check_termination, 1) * 8;
  // certificates (solver.jl:326-349): iteration k ci sets the flag, delta_y is captured at the top of iteration k ci + 1 and the tests run at
  // its end -- the persistent launch is cut at those two iterations (ci = 1 never tests: every iteration re-sets the flag, as in the reference)
  const long long ci = b->prm.check_infeasibility;
  const bool inf_on = ci > 1 && ci < b->prm.max_iter;
  bool captured = false;
  long long target = 0;
  std::vector<BCtl> c(b->nprob);
  int first = 1;
  std::vector<BAa> acc;
  if (b->aa_on) {
    // Accelerated batches: the workgroups take every decision themselves, including WHEN their certificates are due (the first non-accelerated
    // iteration after a flagged one -- different per problem).  A workgroup leaves its launch at that iteration (BAa::need_inf); the host runs
    // k_batch_inf_check on the flagged problems and relaunches towards the same target until every undecided problem has reached it.
    acc.resize((size_t)b->nprob);
    // per launch the host only needs {status, iter} of every problem: a 16-byte record per problem through a pinned buffer (the full BCtl + BAa
    // arrays, 2.8 KB per problem, were copied after EVERY relaunch); the full state is fetched once, after the loop
    if (!b->d_state) {
      BState* ds = nullptr; BHIP(b, hipMalloc((void**)&ds, sizeof(BState) * (size_t)b->nprob)); b->allocs.push_back(ds); b->d_state = ds;
      BHIP(b, hipHostMalloc((void**)&b->h_state, sizeof(BState) * (size_t)b->nprob, hipHostMallocDefault));
    }
    const BState* hs = (const BState*)b->h_state;
    for (;;) {
      target = std::min<long long>(target + slice, b->prm.max_iter);
      bool all = true, timed_out = false;
      for (;;) {
        { const int32_t lrc = launch_batch_admm(b, P, target, first); if (lrc) return lrc; }
        first = 0;
        if (P.check_inf > 0)
          hipLaunchKernelGGL(k_batch_inf_check, dim3(b->nprob), dim3(COSMO_BS), 0, b->stream, b->D, (real)b->prm.eps_prim_inf, (real)b->prm.eps_dual_inf, 1);
        hipLaunchKernelGGL(k_batch_pack_state, dim3((b->nprob + 255) / 256), dim3(256), 0, b->stream, b->D, (BState*)b->d_state);
        BHIP(b, hipMemcpyAsync(b->h_state, b->d_state, sizeof(BState) * (size_t)b->nprob, hipMemcpyDeviceToHost, b->stream));
        BHIP(b, hipStreamSynchronize(b->stream));
        all = true;
        bool behind = false;
        for (int k = 0; k < b->nprob; ++k) if (hs[k].status == 0) { all = false; if (hs[k].iter < target) behind = true; }
        if (b->prm.time_limit != 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > b->prm.time_limit) { timed_out = !all; break; }
        if (!behind) break;
      }
      if (timed_out || all || target >= b->prm.max_iter) {
        BHIP(b, hipMemcpyAsync(c.data(), b->D.ctl, sizeof(BCtl) * b->nprob, hipMemcpyDeviceToHost, b->stream));
        BHIP(b, hipMemcpyAsync(acc.data(), b->D.aa, sizeof(BAa) * b->nprob, hipMemcpyDeviceToHost, b->stream));
        BHIP(b, hipStreamSynchronize(b->stream));
        if (timed_out) for (auto& x : c) if (x.status == 0) x.status = COSMO_HIP_TIME_LIMIT_REACHED;
        break;
      }
    }
  } else
  for (;;) {
    long long stop = captured ? target + 1 : target + slice;
    if (inf_on && !captured) stop = std::min(stop, (target / ci + 1) * ci);
    target = std::min<long long>(stop, b->prm.max_iter);
    { const int32_t lrc = launch_batch_admm(b, P, target, first); if (lrc) return lrc; }
    first = 0;
    if (captured) {
      hipLaunchKernelGGL(k_batch_inf_check, dim3(b->nprob), dim3(COSMO_BS), 0, b->stream, b->D, (real)b->prm.eps_prim_inf, (real)b->prm.eps_dual_inf, 0);
      captured = false;
    } else if (inf_on && target % ci == 0 && target < b->prm.max_iter) {
      hipLaunchKernelGGL(k_batch_inf_capture, dim3(b->nprob), dim3(COSMO_BS), 0, b->stream, b->D);
      captured = true;
    }
    BHIP(b, hipMemcpyAsync(c.data(), b->D.ctl, sizeof(BCtl) * b->nprob, hipMemcpyDeviceToHost, b->stream));
    BHIP(b, hipStreamSynchronize(b->stream));
    bool all = true;
    for (auto& x : c) if (x.status == 0) { all = false; break; }
    if (all || target >= b->prm.max_iter) break;
    if (b->prm.time_limit != 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > b->prm.time_limit) {
      for (auto& x : c) if (x.status == 0) x.status = COSMO_HIP_TIME_LIMIT_REACHED;
      break;
    }
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int k = 0; k < b->nprob; ++k) {
    cosmo_hip_result& r = results[k];
    memset(&r, 0, sizeof r);
    r.status = c[k].status; r.n_rho_updates = c[k].n_rho_updates; r.iter = c[k].iter + (b->aa_on ? acc[(size_t)k].sg_iter : 0); r.safeguarding_iter = b->aa_on ? acc[(size_t)k].sg_iter : 0; r.kkt_iters_total = c[k].kkt_iters_total;
    r.kkt_solves = c[k].solves; r.cost = (double)c[k].cost; r.r_prim = (double)c[k].r_prim; r.r_dual = (double)c[k].r_dual;
    r.max_norm_prim = (double)c[k].max_norm_prim; r.max_norm_dual = (double)c[k].max_norm_dual; r.rho = (double)c[k].rho; r.iter_time = el;
    for (int i = 0; i < COSMO_HIP_MAX_RHO_UPDATES && i < c[k].n_rho_updates; ++i) r.rho_updates[i] = (double)c[k].rho_updates[i];
  }
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// MERGED optimize over `count` one-problem batches of different structure (csrc/batch_group.hip: the singleton classes of a heterogeneous list).  The
// reference's batch mode is `for model in models; optimize!(model); end` (src/solver.jl:78): a list of 256 models of 256 shapes used to be 256 host loops
// of single-workgroup launches; here it is ONE host loop (the non-accelerated branch of cosmo_hip_batch_optimize) whose every launch covers all of them --
// workgroup c runs the streaming kernel on the descriptor of batch c.  Same per-problem arithmetic as k_batch_admm (batch_admm_body), hence the same
// iterates as a one-problem streaming batch; certificates and time limit as there.  Requirements (batch_multi_supported): one problem, no accelerator,
// no PSD cone of side 17..64 (their Jacobi workspace is sized per launch).  All batches carry the group's parameters.
// ---------------------------------------------------------------------------------------------------------------------
bool batch_multi_supported(const cosmo_hip_batch* b) { return b && b->finalized && b->nprob == 1 && !b->aa_on && b->D.nmid == 0 && b->D.A.rowptr != nullptr; }
bool batch_multi_needs_ext(const cosmo_hip_batch* b) { return b->D.npsd > 0 || b->D.n3 > 0; }

int32_t batch_multi_optimize(cosmo_hip_batch** bs, int count, bool ext, cosmo_hip_result* results) {
  if (count <= 0) return COSMO_HIP_OK;
  cosmo_hip_batch* b = bs[0];                                     // errors are reported on the first batch; its stream carries the merged launches
  for (int c = 0; c < count; ++c) if (!bs[c]->have_iterates) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_multi_optimize: set_iterates first");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  for (int c = 0; c < count; ++c) BHIP(b, hipStreamSynchronize(bs[c]->stream));          // every batch's set_iterates ran on its own stream
  std::vector<BatchDev> hd((size_t)count);
  for (int c = 0; c < count; ++c) hd[(size_t)c] = bs[c]->D;
  BatchDev* dD = nullptr; BCtl* dC = nullptr;
  BHIP(b, hipMalloc((void**)&dD, sizeof(BatchDev) * (size_t)count));
  if (hipMalloc((void**)&dC, sizeof(BCtl) * (size_t)count) != hipSuccess) { (void)hipFree(dD); return bfail(b, COSMO_HIP_ERR_HIP, "hipMalloc failed"); }
  auto done = [&](int32_t rc) { (void)hipFree(dD); (void)hipFree(dC); return rc; };
  if (hipMemcpy(dD, hd.data(), sizeof(BatchDev) * (size_t)count, hipMemcpyHostToDevice) != hipSuccess) return done(bfail(b, COSMO_HIP_ERR_HIP, "hipMemcpy failed"));
  const BParams P = bparams(b, true);
  const auto t0 = std::chrono::steady_clock::now();
  const long long slice = std::max<long long>(b->prm.check_termination, 1) * 8;
  const long long ci = b->prm.check_infeasibility;
  const bool inf_on = ci > 1 && ci < b->prm.max_iter;
  bool captured = false;
  long long target = 0;
  std::vector<BCtl> c((size_t)count);
  int first = 1;
  for (;;) {
    long long stop = captured ? target + 1 : target + slice;
    if (inf_on && !captured) stop = std::min(stop, (target / ci + 1) * ci);
    target = std::min<long long>(stop, b->prm.max_iter);
    if (ext) hipLaunchKernelGGL((k_batch_admm_multi<true>), dim3(count), dim3(COSMO_BS), 0, b->stream, (const BatchDev*)dD, P, target, first);
    else hipLaunchKernelGGL((k_batch_admm_multi<false>), dim3(count), dim3(COSMO_BS), 0, b->stream, (const BatchDev*)dD, P, target, first);
    first = 0;
    if (captured) {
      hipLaunchKernelGGL(k_batch_inf_check_multi, dim3(count), dim3(COSMO_BS), 0, b->stream, (const BatchDev*)dD, (real)b->prm.eps_prim_inf, (real)b->prm.eps_dual_inf);
      captured = false;
    } else if (inf_on && target % ci == 0 && target < b->prm.max_iter) {
      hipLaunchKernelGGL(k_batch_inf_capture_multi, dim3(count), dim3(COSMO_BS), 0, b->stream, (const BatchDev*)dD);
      captured = true;
    }
    hipLaunchKernelGGL(k_batch_pack_ctl_multi, dim3((count + 255) / 256), dim3(256), 0, b->stream, (const BatchDev*)dD, count, dC);
    if (hipMemcpyAsync(c.data(), dC, sizeof(BCtl) * (size_t)count, hipMemcpyDeviceToHost, b->stream) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess)
      return done(bfail(b, COSMO_HIP_ERR_HIP, "merged batch launch failed: %s", hipGetErrorString(hipGetLastError())));
    bool all = true;
    for (auto& x : c) if (x.status == 0) { all = false; break; }
    if (all || target >= b->prm.max_iter) break;
    if (b->prm.time_limit != 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > b->prm.time_limit) {
      for (auto& x : c) if (x.status == 0) x.status = COSMO_HIP_TIME_LIMIT_REACHED;
      break;
    }
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int k = 0; k < count; ++k) {
    cosmo_hip_result& r = results[k];
    memset(&r, 0, sizeof r);
    r.status = c[(size_t)k].status; r.n_rho_updates = c[(size_t)k].n_rho_updates; r.iter = c[(size_t)k].iter; r.kkt_iters_total = c[(size_t)k].kkt_iters_total;
    r.kkt_solves = c[(size_t)k].solves; r.cost = (double)c[(size_t)k].cost; r.r_prim = (double)c[(size_t)k].r_prim; r.r_dual = (double)c[(size_t)k].r_dual;
    r.max_norm_prim = (double)c[(size_t)k].max_norm_prim; r.max_norm_dual = (double)c[(size_t)k].max_norm_dual; r.rho = (double)c[(size_t)k].rho; r.iter_time = el;
    for (int i = 0; i < COSMO_HIP_MAX_RHO_UPDATES && i < c[(size_t)k].n_rho_updates; ++i) r.rho_updates[i] = (double)c[(size_t)k].rho_updates[i];
  }
  return done(COSMO_HIP_OK);
}

// Runs exactly n_iters more iterations on every undecided problem (benchmark / parity hook); no early exit on time.
extern "C" int32_t cosmo_hip_batch_iterate(cosmo_hip_batch* b, int64_t n_iters, int32_t with_init) {
  if (!b || !b->have_iterates) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_iterate: set_iterates first");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  const BParams P = bparams(b, false);
  b->iters_done += n_iters;
  { const int32_t lrc = launch_batch_admm(b, P, (long long)b->iters_done, with_init ? 1 : 0); if (lrc) return lrc; }
  BHIP(b, hipStreamSynchronize(b->stream));
  return COSMO_HIP_OK;
}

// which kernel a batch runs (after set_params): out = {form: 0 streaming, 1 LDS image, 2 register kernel <512, 1, 2>, 3 register kernel <512, 2, 4>;
// sliced image 0 / 1; dynamic LDS bytes per workgroup; P held in registers 0 / 1; registers per thread and scratch bytes per thread of that
// instantiation as the loaded code object reports them (hipFuncGetAttributes: VGPRs + AGPRs of the unified file; scratch > 0 = it spills);
// its static LDS bytes; bit 0: length-sorted compute assignment in the Krylov loop, bit 1: cooperative long-row passes (the LONG instantiations)}
extern "C" int32_t cosmo_hip_batch_kernel_info(cosmo_hip_batch* b, int64_t* out) {
  if (!b || !out) return COSMO_HIP_ERR_INVALID;
  if (!b->finalized) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_kernel_info: set_params first");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  { hipFuncAttributes fa;
    BHIP(b, hipFuncGetAttributes(&fa, batch_kernel_of(b).fn));
    out[4] = fa.numRegs; out[5] = (int64_t)fa.localSizeBytes; out[6] = (int64_t)fa.sharedSizeBytes; }
  out[7] = (b->D.permA ? 1 : 0) | ((b->d_img && b->reg_mode != 0 && b->long_rows && !b->D.sliced) ? 2 : 0);
  out[0] = !b->d_img ? 0 : (b->reg_mode == 1 ? 2 : (b->reg_mode == 2 ? 3 : 1));
  out[1] = (b->d_img && b->reg_mode == 1 && b->D.sliced) ? 1 : 0;
  out[2] = b->d_img ? b->lds_bytes : 0;
  out[3] = (out[1] && b->D.pdiag) ? 1 : 0;
  return COSMO_HIP_OK;
}

// per problem {ADMM iterations, KKT solves, Krylov iterations in total}: out holds 3 * nprob entries (measurement: the Krylov work a batch
// step actually did, cf. `iteration_counter` / `multiplications` of kktsolver_indirect.jl:32,56)
extern "C" int32_t cosmo_hip_batch_get_counters(cosmo_hip_batch* b, int64_t* out) {
  if (!b || !b->finalized || !out) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_get_counters: bad call");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  std::vector<BCtl> c(b->nprob);
  BHIP(b, hipMemcpyAsync(c.data(), b->D.ctl, sizeof(BCtl) * b->nprob, hipMemcpyDeviceToHost, b->stream));
  BHIP(b, hipStreamSynchronize(b->stream));
  for (int k = 0; k < b->nprob; ++k) { out[3 * k] = c[k].iter; out[3 * k + 1] = c[k].solves; out[3 * k + 2] = c[k].kkt_iters_total; }
  return COSMO_HIP_OK;
}

// w, w_prev: n+m ; s, mu: m  of problem k
extern "C" int32_t cosmo_hip_batch_get_iterates(cosmo_hip_batch* b, int64_t k, real* w, real* w_prev, real* s, real* mu) {
  if (!b || !b->have_iterates || k < 0 || k >= b->nprob) return bfail(b, COSMO_HIP_ERR_INVALID, "batch_get_iterates: bad call");
  if (hipSetDevice(b->device) != hipSuccess) return bfail(b, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  const size_t N = (size_t)(b->n + b->m), m = (size_t)b->m;
  if (w) BHIP(b, hipMemcpy(w, b->D.w + (size_t)k * N, N * sizeof(real), hipMemcpyDeviceToHost));
  if (w_prev) BHIP(b, hipMemcpy(w_prev, b->D.w_prev + (size_t)k * N, N * sizeof(real), hipMemcpyDeviceToHost));
  if (s) BHIP(b, hipMemcpy(s, b->D.s + (size_t)k * m, m * sizeof(real), hipMemcpyDeviceToHost));
  if (mu) BHIP(b, hipMemcpy(mu, b->D.mu + (size_t)k * m, m * sizeof(real), hipMemcpyDeviceToHost));
  return COSMO_HIP_OK;
}
