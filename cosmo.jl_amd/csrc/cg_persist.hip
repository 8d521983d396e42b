// cg_persist.hip -- the reduced-system CG solve (IterativeSolvers v0.9 cg!, src/linear_solver/kktsolver_indirect.jl:57-70) as ONE
// persistent launch confined to the workgroups of ONE XCD, for operators small enough to live in that XCD's 4 MB L2 (the split CG
// operator of a decomposed SDP, small and medium QPs).
//
// Why: on such operators the four kernels of a Krylov iteration (k_cg_dir, k_spmv_A_rho, k_op_apply, k_cg_upd) take 3-5 us each,
// i.e. they are pure launch / dependent-load latency, and the ~3.5 us the host needs per launch makes the whole solve HOST-bound
// (BASELINE config 5: 170-210 Krylov iterations per ADMM iteration, 4 launches each; profiles/r02_cfg5_kernel_stats_v1.csv).
// Here the iteration is a loop inside one kernel with three software barriers:
//
//   [res_k = ||r||, stop test, beta]                         from the r'r partials           (every workgroup, redundantly)
//   u_k = r + beta u_{k-1} (owned elements, stored for later gathers)  and  tmp = rho .* (A u_k) with u_k RECOMPUTED at the
//   gathered columns from r and u_{k-1} (same expression, same bits: no barrier between the direction update and the product)
//   ---- barrier ----   c = [P | A'] [u_k; tmp] + sigma u_k (+ diag .* u_k), partials of u'c
//   ---- barrier ----   alpha ; x += alpha u ; r -= alpha c ; partials of r'r
//   ---- barrier ----
//
// Bit-exactness with the multi-kernel path (asserted in tests/test_gpu_cg_persist.py): a 1024-thread workgroup is four QUARTERS of
// 256 threads, and a quarter executes the arithmetic of one 256-thread workgroup of the original kernels -- one CSR-stream tile
// (same left-to-right row sums from LDS) or one block of 256 vector elements -- including its block reduction tree (wave butterfly,
// then the four wave sums in order); the partial arrays have the same length and order, and every workgroup folds them with the
// same strided sums as the original consumer kernels.
//
// Inter-workgroup visibility WITHOUT fences: all participants run on the same XCD (each candidate block reads HW_REG_XCC_ID and
// only blocks of XCD 0 take a ticket; placement is verified at run time, not assumed), so the XCD's L2 is their coherence point.
// Producers use plain stores (write-through L1, the line stays in L2) followed by `s_waitcnt vmcnt(0)`; consumers read every
// mutable vector with sc1 loads (relaxed agent-scope atomic loads: L1 bypass, L2-served).  No buffer_wbl2 / buffer_inv (1.7 us
// each).  The barrier is a monotonic arrival counter polled with sc1 loads.  Every spin is bounded: a start-up rendezvous that
// does not complete (fewer than W blocks of the launch landed on XCD 0, or they were not co-resident) leaves all data untouched and
// reports a Krylov "stall", which the host resolves with the multi-kernel path and then stops using the persistent kernel.
#include "device_utils.h"

#define PCG_Q 4                          // quarters (virtual 256-thread workgroups) per workgroup
#define PCG_THREADS (COSMO_BS * PCG_Q)   // 1024
#define PCG_SPIN_LIMIT (1L << 22)

struct PcgArgs {
  CsrView A, PT;                 // operator pieces (Am / [P | Am'] when the operator is split)
  const real* rho;             // rho on A's rows
  const real* diag;            // diagonal part of A' rho A (split operator) or null
  real sigma;
  real *x, *r, *c, *tmp, *u0, *u1;
  real *part_rr, *part_uc;
  int n_rr0;                     // number of r'r partials left by the solve-start kernel
  int nvec;                      // vector blocks of 256 elements
  long long n, maxiter;
  unsigned* sync;                // [0] tickets, [1] barrier arrivals, [2] abort
  int W;                         // participating workgroups
  int cap;                       // LDS doubles per quarter (>= the largest tile)
  int guard;
};

__device__ __forceinline__ real ld2(const real* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// indexed form with an UNSIGNED 32-bit index: lets the backend use the scalar-base + 32-bit vector-offset addressing mode instead of
// keeping a 64-bit VGPR address per gather alive (the kernel is VGPR-bound)
__device__ __forceinline__ real ld2i(const real* base, unsigned idx) {
  asm volatile("" : "+v"(idx));        // opaque: the 64-bit address is rebuilt at the use (2 VALU ops) instead of being hoisted out of the Krylov loop
  return __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned opaque(int idx) { unsigned u = (unsigned)idx; asm volatile("" : "+v"(u)); return u; }
__device__ __forceinline__ unsigned ld2u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block_sum of device_utils.h for the 256 threads of one quarter (tq = thread index in the quarter, red_q = 4 doubles of the quarter)
__device__ __forceinline__ real q_sum(real v, real* red_q, int tq) {
  v = wave_sum(v);
  __syncthreads();
  if ((tq & 63) == 0) red_q[tq >> 6] = v;
  __syncthreads();
  real t = 0.0;
#pragma unroll
  for (int i = 0; i < COSMO_BS / 64; ++i) t += red_q[i];
  return t;
}
// reduce_partials_sum of device_utils.h; the partials were written by other workgroups: sc1 loads
__device__ __forceinline__ real q_reduce_partials(const real* p, int count, real* red_q, int tq) {
  real a = 0.0;
  for (int i = tq; i < count; i += COSMO_BS) a += ld2(p + i);
  return q_sum(a, red_q, tq);
}

// arrival counter barrier among the W participants (same XCD: the L2 is the coherence point, see the header)
__device__ __forceinline__ bool pcg_barrier(unsigned* sync, unsigned target, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (ld2u(sync + 1) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > PCG_SPIN_LIMIT || ld2u(sync + 2) != 0u) { __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_flag = 1; break; }
    }
  }
  __syncthreads();
  return *s_flag == 0;
}

// Work of one quarter, fixed for the whole solve and held in REGISTERS: at most PCG_TA tiles of A, PCG_TP tiles of [P | A'] and
// PCG_TV blocks of 256 vector elements; every tile has at most 256 nonzeros and 256 rows (the size-adaptive CSR-stream schedule of
// small operators), so a thread owns at most ONE nonzero and ONE row of each of its tiles -- exactly what thread tq of the
// original 256-thread workgroup owns.
#define PCG_TILE 256

// sum of the four wave sums in order, for NT independent reductions at once (one barrier pair): the tree of block_sum
template <int NT>
__device__ __forceinline__ void q_sum_multi(real (&v)[NT], real* red_q, int tq) {
#pragma unroll
  for (int j = 0; j < NT; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if ((tq & 63) == 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) red_q[4 * j + (tq >> 6)] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    real t = 0.0;
#pragma unroll
    for (int i = 0; i < COSMO_BS / 64; ++i) t += red_q[4 * j + i];
    v[j] = t;
  }
}

template <int PCG_TA, int PCG_TP, int PCG_TV>
__global__ __launch_bounds__(PCG_THREADS) void k_cg_persist(Ctl* __restrict__ ctl, PcgArgs a) {
  // LDS: per quarter PCG_TP * PCG_TILE staged products and 4 * PCG_TP reduction slots; then 2 ints
  constexpr int PCG_TS = PCG_TA > PCG_TP ? PCG_TA : PCG_TP;
  __shared__ real s_stage[PCG_Q][PCG_TS][PCG_TILE];
  __shared__ real s_red[PCG_Q][4 * (PCG_TS > PCG_TV ? PCG_TS : PCG_TV)];
  // per-thread constants of the solve (matrix values, rho, diag) live in LDS, not in VGPRs: [slot][thread], conflict-free
  __shared__ real s_aval[PCG_TA][PCG_THREADS], s_pval[PCG_TP][PCG_THREADS];     // rho / diag of the owned rows: plain (L1-cached) loads
  __shared__ int s_int[2];                 // [0] participant rank, [1] abort flag
  if (threadIdx.x == 0) {
    int rank = -1;
    const bool skip = (a.guard && ctl->halt) || ctl->cg_done;
    if (!skip) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      if ((xcc & 0xFu) == 0u) {
        const unsigned t = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < (unsigned)a.W) rank = (int)t;
      }
    }
    s_int[0] = rank; s_int[1] = 0;
  }
  __syncthreads();
  const int wg = s_int[0];
  if (wg < 0) return;
  // start-up rendezvous: all W participants hold a ticket (nothing has been modified yet if this fails)
  if (threadIdx.x == 0) {
    long spins = 0;
    while (ld2u(a.sync) < (unsigned)a.W) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > PCG_SPIN_LIMIT / 8 || ld2u(a.sync + 2) != 0u) { __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_int[1] = 1; break; }
    }
  }
  __syncthreads();
  if (s_int[1]) {
    if (wg == 0 && threadIdx.x == 0) { ctl->stalled = 1; ctl->halt = 1; }      // resolved by the host with the multi-kernel path
    return;
  }
  const int q = threadIdx.x >> 8, tq = threadIdx.x & 255;
  real* red_q = s_red[q];
  const int vq = wg * PCG_Q + q;                 // virtual workgroup id of this quarter
  const int nq = a.W * PCG_Q;
  const long long n = a.n;
  const real tol = ctl->tol;
  const real sigma = a.sigma;

  // ---- the quarter's share of the operator, loaded once (rho / diag are constant during a solve) -------------------------
  // Clamped indices: EVERY load of the loop is unconditional (an absent nonzero / row / element reads entry 0 and is discarded by a
  // select or multiplied by a zero matrix value).  A load under a per-thread `if` makes hipcc branch around it and wait vmcnt(0) per
  // element -- the phases would be chains of serialized L2 round trips (measured: 37 us per iteration instead of 7).
  unsigned acolc[PCG_TA], arowc[PCG_TA], pcolc[PCG_TP], prowc[PCG_TP], vic[PCG_TV];
  int aoff[PCG_TA];         // lo | hi << 16: offsets of the row's products in the tile (<= 256)
  int poff[PCG_TP];         // lo | split << 10 | hi << 20
  unsigned valid = 0u;      // bit j: A row j owned; bit 8 + j: [P | A'] row j owned; bit 16 + j: vector element j owned
#pragma unroll
  for (int j = 0; j < PCG_TA; ++j) {
    acolc[j] = 0u; arowc[j] = 0u; aoff[j] = 0;
    real aval = 0.0;
    const int tile = vq + j * nq;
    if (tile < a.A.nb) {
      const int4 d = reinterpret_cast<const int4*>(a.A.rb)[tile];
      if (tq < d.w - d.z) { acolc[j] = (unsigned)a.A.col[d.z + tq]; aval = a.A.val[d.z + tq]; }
      const int r = d.x + tq;
      if (r < d.y) { arowc[j] = (unsigned)r; valid |= 1u << j; aoff[j] = (a.A.rowptr[r] - d.z) | ((a.A.rowptr[r + 1] - d.z) << 16); }
    }
    s_aval[j][threadIdx.x] = aval;
  }
#pragma unroll
  for (int j = 0; j < PCG_TP; ++j) {
    pcolc[j] = 0u; prowc[j] = 0u; poff[j] = 0;
    real pval = 0.0;
    const int tile = vq + j * nq;
    if (tile < a.PT.nb) {
      const int4 d = reinterpret_cast<const int4*>(a.PT.rb)[tile];
      if (tq < d.w - d.z) { pcolc[j] = (unsigned)a.PT.col[d.z + tq]; pval = a.PT.val[d.z + tq]; }
      const int r = d.x + tq;
      if (r < d.y) {
        const int lo = a.PT.rowptr[r] - d.z, hi = a.PT.rowptr[r + 1] - d.z;
        const int sp = a.PT.split ? (a.PT.split[r] - d.z) : hi;
        prowc[j] = (unsigned)r; valid |= 1u << (8 + j); poff[j] = lo | (sp << 10) | (hi << 20);
      }
    }
    s_pval[j][threadIdx.x] = pval;
  }
  // owned vector elements: x, r, u live in registers for the whole solve (r and u are also stored for the other workgroups' gathers)
  real vx[PCG_TV], vr[PCG_TV], vu[PCG_TV];
#pragma unroll
  for (int j = 0; j < PCG_TV; ++j) {
    const int vb = vq + j * nq;
    const long long i0 = (long long)vb * COSMO_BS + tq;
    vic[j] = 0u; vx[j] = 0.0; vr[j] = 0.0; vu[j] = 0.0;
    if (vb < a.nvec && i0 < n) { vic[j] = (unsigned)i0; valid |= 1u << (16 + j); vx[j] = a.x[i0]; vr[j] = a.r[i0]; }
  }
  real* uold = a.u0;
  real* unew = a.u1;
  real prev = 1.0;
  int n_rr = a.n_rr0;
  unsigned bar = 0;
  int k = 0;
#ifdef PCG_TIMING
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#define PCG_MARK(i) { const unsigned long long tn_ = wall_clock64(); tph[i] += tn_ - tlast; tlast = tn_; }
#else
#define PCG_MARK(i)
#endif
  const int split_col = a.PT.split_col;
  const bool has_diag = a.diag != nullptr;
  const real* diagp = has_diag ? a.diag : a.rho;      // a valid address either way; the value is discarded when there is no diagonal
  for (;; ++k) {
    // ---- loads of the first phase, all in flight together: r'r partials, gathers of r and u_{k-1}, rho of the owned rows ----------
    real gr[PCG_TA], gu[PCG_TA], rh[PCG_TA];
#pragma unroll
    for (int j = 0; j < PCG_TA; ++j) { gr[j] = ld2i(a.r, acolc[j]); gu[j] = ld2i(uold, acolc[j]); rh[j] = a.rho[arowc[j]]; }
    real pa = 0.0;
    for (int i = tq; i < n_rr; i += COSMO_BS) pa += ld2(a.part_rr + i);
    // ---- k_cg_dir: residual norm, stopping rule (checked BEFORE the iteration), beta --------------------------------
    const real rr = q_sum(pa, red_q, tq);
    const real res = sqrt(rr);
    const bool done = ((long long)k >= a.maxiter) || (res <= tol);
    if (wg == 0 && threadIdx.x == 0) {
      ctl->resv[k & 1] = res;
      if (done) { ctl->cg_done = 1; ctl->cg_k = k; }
    }
    if (done) break;
    const real beta = (res * res) / (prev * prev);
    PCG_MARK(0)
    // direction update on the owned elements (stored for the gathers of the operator kernel and of the next iteration)
#pragma unroll
    for (int j = 0; j < PCG_TV; ++j) {
      vu[j] = vr[j] + beta * vu[j];                                          // k == 0: vu = 0, i.e. r + beta * 0.0 as k_cg_dir
      if (valid & (1u << (16 + j))) unew[vic[j]] = vu[j];
    }
    // ---- k_spmv_A_rho: tmp = rho .* (A u), u recomputed at the gathered columns (same expression as the owner's) -------
#pragma unroll
    for (int j = 0; j < PCG_TA; ++j) {
      const real u0 = (k > 0) ? gu[j] : 0.0;
      s_stage[q][j][tq] = s_aval[j][threadIdx.x] * (gr[j] + beta * u0);       // absent nonzero: value 0 in a slot no row reads
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PCG_TA; ++j) {
      real s1 = 0.0, s2 = 0.0;
      for (int e = aoff[j] & 0xFFFF; e < (aoff[j] >> 16); ++e) s1 += s_stage[q][j][e];
      if (valid & (1u << j)) a.tmp[arowc[j]] = (s1 + s2) * rh[j];
    }
    PCG_MARK(1)
    bar += (unsigned)a.W;
    if (!pcg_barrier(a.sync, bar, s_int + 1)) break;
    PCG_MARK(2)
    // ---- k_op_apply (mode 1): c = P u + (sigma u + A' tmp) [+ diag .* u], partials of u'c -----------------------------
    real gp[PCG_TP], gv[PCG_TP], dg[PCG_TP], acc[PCG_TP];
#pragma unroll
    for (int j = 0; j < PCG_TP; ++j) {
      const bool left = (int)pcolc[j] < split_col;
      const real* base = left ? unew : a.tmp;
      const unsigned idx = left ? pcolc[j] : pcolc[j] - (unsigned)split_col;
      gp[j] = ld2i(base, idx);
      gv[j] = ld2i(unew, prowc[j]);
      dg[j] = diagp[prowc[j]];
    }
#pragma unroll
    for (int j = 0; j < PCG_TP; ++j) s_stage[q][j][tq] = s_pval[j][threadIdx.x] * gp[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PCG_TP; ++j) {
      real s1 = 0.0, s2 = 0.0;
      const int lo = poff[j] & 0x3FF, sp = (poff[j] >> 10) & 0x3FF, hi = poff[j] >> 20;
      for (int e = lo; e < sp; ++e) s1 += s_stage[q][j][e];
      for (int e = sp; e < hi; ++e) s2 += s_stage[q][j][e];
      const real vj = gv[j];
      real cj = s1 + (sigma * vj + s2);
      const real cd = cj + dg[j] * vj;
      cj = has_diag ? cd : cj;
      if (valid & (1u << (8 + j))) a.c[prowc[j]] = cj;
      const real t = R(0.0) + vj * cj;
      acc[j] = (valid & (1u << (8 + j))) ? t : 0.0;
    }
    q_sum_multi<PCG_TP>(acc, red_q, tq);
    if (tq == 0) {
#pragma unroll
      for (int j = 0; j < PCG_TP; ++j) if (vq + j * nq < a.PT.nb) a.part_uc[vq + j * nq] = acc[j];
    }
    PCG_MARK(3)
    bar += (unsigned)a.W;
    if (!pcg_barrier(a.sync, bar, s_int + 1)) break;
    PCG_MARK(4)
    // ---- k_cg_upd: alpha ; x += alpha u ; r -= alpha c ; partials of r'r ------------------------------------------------
    real c0[PCG_TV];
#pragma unroll
    for (int j = 0; j < PCG_TV; ++j) c0[j] = ld2i(a.c, vic[j]);
    real pu = 0.0;
    for (int i = tq; i < a.PT.nb; i += COSMO_BS) pu += ld2(a.part_uc + i);
    const real uc = q_sum(pu, red_q, tq);
    const real alpha = (res * res) / uc;
    real racc[PCG_TV];
#pragma unroll
    for (int j = 0; j < PCG_TV; ++j) {
      vx[j] = vx[j] + alpha * vu[j];
      const real ri = vr[j] - alpha * c0[j];
      vr[j] = (valid & (1u << (16 + j))) ? ri : 0.0;
      if (valid & (1u << (16 + j))) a.r[vic[j]] = ri;
      const real t = R(0.0) + ri * ri;
      racc[j] = (valid & (1u << (16 + j))) ? t : 0.0;
    }
    q_sum_multi<PCG_TV>(racc, red_q, tq);
    if (tq == 0) {
#pragma unroll
      for (int j = 0; j < PCG_TV; ++j) if (vq + j * nq < a.nvec) a.part_rr[vq + j * nq] = racc[j];
    }
    n_rr = a.nvec;
    PCG_MARK(5)
    bar += (unsigned)a.W;
    if (!pcg_barrier(a.sync, bar, s_int + 1)) break;
    PCG_MARK(6)
    prev = res;
    real* t = uold; uold = unew; unew = t;
  }
  // the iterate (register-resident during the solve) and the last direction (the multi-kernel path may resume from them)
#pragma unroll
  for (int j = 0; j < PCG_TV; ++j) if (valid & (1u << (16 + j))) { a.x[vic[j]] = vx[j]; if (uold != a.u0) a.u0[vic[j]] = vu[j]; }
  if (s_int[1] && wg == 0 && threadIdx.x == 0) ctl->error = COSMO_HIP_ERR_HIP;     // a mid-solve barrier timed out: unrecoverable
#ifdef PCG_TIMING
  if (wg == 0 && threadIdx.x == 0) { for (int i = 0; i < 7; ++i) a.sync[8 + i] += (unsigned)tph[i]; a.sync[15] += (unsigned)k; }
#endif
}

__global__ void k_pcg_reset(unsigned* sync) { sync[0] = 0u; sync[1] = 0u; sync[2] = 0u; sync[3] = 0u; }

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// instantiated (TA, TP, TV) combinations, smallest first
#define PCG_NVARIANTS 2
static const int kPcgVariants[PCG_NVARIANTS][3] = {{2, 4, 2}, {4, 6, 2}};

// Decides whether the handle's CG operator qualifies and prepares the launch; called from set_params after build_op_split.
int32_t pcg_setup(cosmo_hip_handle* h) {
  h->pcg_on = false;
  if (h->pcg_sync) { (void)hipFree(h->pcg_sync); h->pcg_sync = nullptr; }
  if (h->pcg_u2) { (void)hipFree(h->pcg_u2); h->pcg_u2 = nullptr; }
  // OPT-IN (COSMO_HIP_CG_PERSIST=1).  Measured on MI355X (profiles/r02_cg_persist.md): bit-identical to the multi-kernel loop, but not
  // faster -- 18-46 us per Krylov iteration against 16-22 us for four launches: one XCD's L2 has to serve every gather of the
  // iteration as an uncached (sc1) 8-byte request, and three software barriers cost about what the four kernel boundaries cost.
  int want = 0;
  if (const char* e = getenv("COSMO_HIP_CG_PERSIST")) want = atoi(e) ? 1 : 0;
  if (want == 0 || h->prm.kkt_kind != COSMO_HIP_KKT_CG || h->cg_sr || h->n == 0) return COSMO_HIP_OK;
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  hipDeviceProp_t prop;
  HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
  const int W = std::max(1, prop.multiProcessorCount / 8);             // one 1024-thread workgroup per CU of one XCD
  const int nq = W * PCG_Q;
  const long long nvec = (h->n + COSMO_BS - 1) / COSMO_BS;
  // hard limits: the quarter's share of the operator is register-resident (TA / TP tiles, TV vector blocks), one partial per tile /
  // vector block as in the multi-kernel path, tiles of at most 256 nonzeros and 256 rows (one nonzero and one row per thread)
  int variant = -1;
  for (int v = 0; v < PCG_NVARIANTS; ++v)
    if (Ao.nb <= kPcgVariants[v][0] * nq && PTo.nb <= kPcgVariants[v][1] * nq && nvec <= kPcgVariants[v][2] * nq) { variant = v; break; }
  if (variant < 0) return COSMO_HIP_OK;
  if (PTo.grid != PTo.nb || Ao.grid != std::max(Ao.nb, 1) || PTo.nb > COSMO_MAX_PARTIALS) return COSMO_HIP_OK;
  for (const CsrDev* M : {&Ao, &PTo}) {
    std::vector<int> rb((size_t)4 * std::max(M->nb, 1));
    HIPCHK(h, hipMemcpy(rb.data(), M->rb, sizeof(int) * rb.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < M->nb; ++k)
      if (rb[4 * k + 3] - rb[4 * k + 2] > PCG_TILE || rb[4 * k + 1] - rb[4 * k] > COSMO_BS) return COSMO_HIP_OK;
  }
  h->pcg_W = W;
  h->pcg_cap = variant;
  h->pcg_smem = 0;
  HIPCHK(h, hipMalloc((void**)&h->pcg_sync, 16 * sizeof(unsigned)));
  HIPCHK(h, hipMalloc((void**)&h->pcg_u2, sizeof(real) * (size_t)std::max<long long>(h->n, 1)));
  HIPCHK(h, hipMemset(h->pcg_sync, 0, 16 * sizeof(unsigned)));
  HIPCHK(h, hipMemset(h->pcg_u2, 0, sizeof(real) * (size_t)std::max<long long>(h->n, 1)));
  h->pcg_on = true;
  return COSMO_HIP_OK;
}

void pcg_free(cosmo_hip_handle* h) {
  if (h->pcg_sync) { (void)hipFree(h->pcg_sync); h->pcg_sync = nullptr; }
  if (h->pcg_u2) { (void)hipFree(h->pcg_u2); h->pcg_u2 = nullptr; }
  h->pcg_on = false;
}

// the whole Krylov loop of one solve (after enqueue_cg_start); returns without synchronising
int32_t pcg_enqueue_solve(cosmo_hip_handle* h, int guard) {
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  PcgArgs a;
  a.A = view_of(Ao); a.PT = view_of(PTo);
  a.rho = h->op_split ? h->op_rho_m : h->rho;
  a.diag = h->op_split ? h->op_diag : nullptr;
  a.sigma = h->prm.sigma;
  a.x = h->x_tl; a.r = h->r; a.c = h->c; a.tmp = h->tmp_m; a.u0 = h->u; a.u1 = h->pcg_u2;
  a.part_rr = h->partials + (size_t)SLOT_RR * COSMO_MAX_PARTIALS;
  a.part_uc = h->partials + (size_t)SLOT_UC * COSMO_MAX_PARTIALS;
  a.n_rr0 = PTo.grid;
  a.nvec = (int)((h->n + COSMO_BS - 1) / COSMO_BS);
  a.n = h->n; a.maxiter = h->n;
  a.sync = h->pcg_sync; a.W = h->pcg_W; a.cap = h->pcg_cap; a.guard = guard;
  hipLaunchKernelGGL(k_pcg_reset, dim3(1), dim3(1), 0, h->stream, h->pcg_sync);
  prof_begin(h, KC_OP_APPLY);
  // candidates: twice as many blocks per XCD as participants are needed (block b is observed on XCD b % 8; only blocks that READ
  // XCC_ID == 0 take part, the surplus returns at once)
  const dim3 G(8 * 2 * h->pcg_W), B(PCG_THREADS);
  if (h->pcg_cap == 0) hipLaunchKernelGGL((k_cg_persist<2, 4, 2>), G, B, 0, h->stream, h->ctl, a);
  else hipLaunchKernelGGL((k_cg_persist<4, 6, 2>), G, B, 0, h->stream, h->ctl, a);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  h->pcg_launches += 1;
  return COSMO_HIP_OK;
}
