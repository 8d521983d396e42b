// anderson.hip -- Anderson acceleration of the ADMM fixed-point iteration on the device: the reference's default
// AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory, NoRegularizer}(mem = 15) with safeguarding
// (src/settings.jl:136-138, src/accelerator_interface.jl:58-130).  The accelerator itself lives in the external package
// COSMOAccelerators.jl (Project.toml:8,27), which is not part of the reference tree: this file restates the published
// algorithm exactly as the CPU test oracle (class AndersonAccelerator) does (PARITY UNPINNED -- see DESIGN.md section 7).
//
//   update!(g = w, x = w_prev):  f = x - g; G[:, j] = g - g_last; v = f - f_last; modified Gram-Schmidt of v against
//                                Q[:, 0..j) -> R[0..j, j], Q[:, j]
//   accelerate!(g = w):          eta = R \ (Q' f);  w -= G eta  unless R is singular or ||eta||_2 > 1e4
//   safeguard:                   decline the candidate if ||w_prev - w|| after the step > tau * ||f||
//
// Layout: G and Q are (n+m) x mem column-major slabs in HBM (cfg4: 2 x 4.0M x 15 doubles = 0.96 GB of the 288 GB); every
// pass over them is a coalesced stream.  The Gram-Schmidt sweep needs one global reduction per column; it is organised
// so that ONE kernel per column does "v -= r_i Q_i" for the r_i reduced from the previous kernel's partials and, in the same
// pass, the partial dots <Q_{i+1}, v>.  Reductions are fixed-order (wave butterfly -> LDS -> every consumer re-reduces
// the producer's partials), so accelerated runs are bit-reproducible.  All data-dependent decisions (success of the least
// squares step, safeguarding) are taken on the device and read back with the one synchronisation per iteration that the
// accelerated loop needs anyway.
//
// Round 6: the non-default variants the reference documents (docs/src/acceleration.md:23-26) -- Type1 and Type2{NormalEquations} with RestartedMemory or
// RollingMemory: histories X (Type1), F, G; the mem x mem matrix M = L' F (L = X resp. F) is kept current with 2 l inner products per update
// (column j and row j), eta = M \ (L' f) by LU with partial pivoting in one workgroup (k_aa_prep_ne, k_aa_gram, k_aa_gram_store, k_aa_solve_ne).
// Restated from the published algorithm like the default variant (PARITY UNPINNED); checked against oracle.AndersonAcceleratorNE.
//
// ROW-SHARDED handles (round 4; csrc/rowshard.hip): w = [x (replicated, n) ; w_s (this rank's rows)], so every inner product of the accelerator is
// (the x-part, counted ONCE: ranks other than 0 leave it out of their partial sums) + the sum over the ranks of the local row parts.  The
// workgroup partials of every reduction are summed over the ranks slot by slot with ONE all-reduce of the partial array (comm_allreduce_sum, the
// exchange path of the loop), then every consumer reduces the summed partials in the usual fixed order: all ranks obtain the SAME bits for R, eta,
// the success flag and the safeguarding decision, so their control flow cannot diverge.  Up to mem + 3 small all-reduces per accelerated iteration.
#include "device_utils.h"
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define AA_MAX_MEM 32

struct AaFlags {
  int success;        // CA.was_successful
  int declined;       // safeguarding rejected the candidate
  int fail_eta, fail_singular;
  real nrm_f;       // ||f|| of the last update! (the non-accelerated fixed-point residual)
  real nrm_f_acc;   // ||w_prev - w|| after the accelerated step
  real eta_norm;
  real R[AA_MAX_MEM * AA_MAX_MEM];   // column-major mem x mem
  real eta[AA_MAX_MEM];
};

struct AaState {
  cosmo_hip_accel_params prm;
  long long N = 0;
  int mem = 0;
  real *G = nullptr, *Q = nullptr, *f = nullptr, *f_last = nullptr, *g_last = nullptr;
  // normal-equations variants (Type1 / Type2{NormalEquations}; kind >= 2): Q holds F (the f-differences), X the x-differences (Type1 only);
  // the mem x mem matrix M = L' F (L = X or F) lives in flags->R and is kept current column by column / row by row
  real *X = nullptr, *x_last = nullptr;
  bool ne = false, type1 = false, rolling = false;
  real* parts = nullptr;      // (2 AA_MAX_MEM + 2) x COSMO_MAX_PARTIALS (the second half: the row partials of the normal-equations variants)
  AaFlags* flags = nullptr;     // device
  AaFlags* flags_host = nullptr;  // pinned
  int grid = 1;
  int nparts = 1;             // partials a consumer reduces: grid, or COSMO_MAX_PARTIALS on a row-sharded handle (the all-reduced array, zero beyond grid)
  long long dot_lo = 0;       // first element that enters this rank's inner products (n on ranks > 0 of a row-sharded run: the x-part is replicated)
  bool sharded = false;
  // host-tracked (data independent) state of the accelerator
  int iter = 0;
  bool init_phase = true;
  bool active = false;
  long long num_accelerated = 0, num_restarts = 0, num_declined = 0, num_accepted = 0;
  long long num_rho_restarts = 0;     // CA.restart! calls because rho was adapted (src/solver.jl:272-275)
};

namespace {

#define AA_PARTS(S, k) ((S)->parts + (size_t)(k) * COSMO_MAX_PARTIALS)

// f = x - g ; first call after a restart: remember (g, f) ; otherwise G_j = g - g_last, v = f - f_last (stored in Q_j) and the
// partial dots <Q_0, v> (or ||v||^2 when j == 0)
__global__ __launch_bounds__(COSMO_BS) void k_aa_prep(long long N, const real* __restrict__ g, const real* __restrict__ x, int init,
                                                      real* __restrict__ f, real* __restrict__ f_last, real* __restrict__ g_last,
                                                      real* __restrict__ Gj, real* __restrict__ v, const real* __restrict__ Q0, int j,
                                                      real* __restrict__ p_out, long long dot_lo) {
  __shared__ real red[COSMO_BS / 64];
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real gi = g[i];
    const real fi = x[i] - gi;
    f[i] = fi;
    if (!init) {
      Gj[i] = gi - g_last[i];
      const real vi = fi - f_last[i];
      v[i] = vi;
      if (i >= dot_lo) acc += (j == 0) ? vi * vi : Q0[i] * vi;
    }
    g_last[i] = gi;
    f_last[i] = fi;
  }
  if (!init) {
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) p_out[blockIdx.x] = acc;
  }
}

// Gram-Schmidt step i (< j): r = sum(p_in) ; R[i, j] = r ; v -= r Q_i ; partial <Q_{i+1}, v> (i + 1 < j) or ||v||^2 (i + 1 == j)
__global__ __launch_bounds__(COSMO_BS) void k_aa_mgs(long long N, int nparts, const real* __restrict__ p_in, const real* __restrict__ Qi,
                                                     const real* __restrict__ Qnext, real* __restrict__ v, int last, real* __restrict__ Rij,
                                                     real* __restrict__ p_out, long long dot_lo) {
  __shared__ real red[COSMO_BS / 64];
  const real r = reduce_partials_sum(p_in, nparts, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) *Rij = r;
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real vi = v[i] - r * Qi[i];
    v[i] = vi;
    if (i >= dot_lo) acc += last ? vi * vi : Qnext[i] * vi;
  }
  __syncthreads();
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) p_out[blockIdx.x] = acc;
}

// R[j, j] = ||v|| ; Q_j = v / ||v||
__global__ __launch_bounds__(COSMO_BS) void k_aa_normalize(long long N, int nparts, const real* __restrict__ p_in, real* __restrict__ v,
                                                           real* __restrict__ Rjj) {
  __shared__ real red[COSMO_BS / 64];
  const real nv = sqrt(reduce_partials_sum(p_in, nparts, red));
  if (blockIdx.x == 0 && threadIdx.x == 0) *Rjj = nv;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) v[i] = v[i] / nv;
}

// partial Q[:, 0..l)' f and ||f||^2
template <int L>
__global__ __launch_bounds__(COSMO_BS) void k_aa_qtf(long long N, int l, const real* __restrict__ Q, const real* __restrict__ f,
                                                     real* __restrict__ parts, long long dot_lo) {
  __shared__ real red[COSMO_BS / 64];
  real acc[L + 1];
#pragma unroll
  for (int k = 0; k <= L; ++k) acc[k] = 0.0;
  for (long long i = dot_lo + (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real fi = f[i];
#pragma unroll
    for (int k = 0; k < L; ++k) if (k < l) acc[k] += Q[(size_t)k * N + i] * fi;
    acc[L] += fi * fi;
  }
#pragma unroll
  for (int k = 0; k <= L; ++k) {
    const int slot = (k == L) ? AA_MAX_MEM : k;
    if (k < l || k == L) {
      const real s = block_sum(acc[k], red);
      if (threadIdx.x == 0) parts[(size_t)slot * COSMO_MAX_PARTIALS + blockIdx.x] = s;
    }
  }
}

// one workgroup: eta = R[0..l, 0..l) \ (Q' f) by back substitution ; success unless singular / non-finite / ||eta|| > eta_max
__global__ __launch_bounds__(COSMO_BS) void k_aa_solve(int l, int nparts, const real* __restrict__ parts, real eta_max, AaFlags* __restrict__ F) {
  __shared__ real red[COSMO_BS / 64];
  __shared__ real rhs[AA_MAX_MEM];
  for (int k = 0; k < l; ++k) {
    const real s = reduce_partials_sum(parts + (size_t)k * COSMO_MAX_PARTIALS, nparts, red);
    if (threadIdx.x == 0) rhs[k] = s;
    __syncthreads();
  }
  const real ff = reduce_partials_sum(parts + (size_t)AA_MAX_MEM * COSMO_MAX_PARTIALS, nparts, red);
  if (threadIdx.x == 0) {
    F->nrm_f = sqrt(ff);
    bool ok = true;
    for (int c = 0; c < l && ok; ++c)
      for (int r = 0; r <= c; ++r) { const real x = F->R[c * AA_MAX_MEM + r]; if (!(fabs(x) <= REAL_MAX)) ok = false; }
    for (int k = 0; k < l; ++k) if (F->R[k * AA_MAX_MEM + k] == R(0.0)) ok = false;
    if (!ok) { F->success = 0; F->fail_singular += 1; return; }
    real nrm2 = 0.0;
    for (int i = l - 1; i >= 0; --i) {
      real s = rhs[i];
      for (int k = i + 1; k < l; ++k) s -= F->R[k * AA_MAX_MEM + i] * F->eta[k];
      const real e = s / F->R[i * AA_MAX_MEM + i];
      F->eta[i] = e;
      nrm2 += e * e;
    }
    const real en = sqrt(nrm2);
    F->eta_norm = en;
    if (!(en <= eta_max)) { F->success = 0; F->fail_eta += 1; return; }     // also catches NaN
    F->success = 1;
  }
}

// w -= G[:, 0..l) eta  when the least-squares step succeeded
template <int L>
__global__ __launch_bounds__(COSMO_BS) void k_aa_apply(long long N, int l, const real* __restrict__ G, const AaFlags* __restrict__ F,
                                                       real* __restrict__ w) {
  if (!F->success) return;
  real e[L];
#pragma unroll
  for (int k = 0; k < L; ++k) e[k] = (k < l) ? F->eta[k] : 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    real s = 0.0;
#pragma unroll
    for (int k = 0; k < L; ++k) if (k < l) s += G[(size_t)k * N + i] * e[k];
    w[i] = w[i] - s;
  }
}

// f = w_prev - w and its partial squared norm (compute_accelerated_res_norm!, accelerator_interface.jl:123-126)
__global__ __launch_bounds__(COSMO_BS) void k_aa_resnorm(long long N, const real* __restrict__ w, const real* __restrict__ w_prev,
                                                         real* __restrict__ f, real* __restrict__ p_out, long long dot_lo) {
  __shared__ real red[COSMO_BS / 64];
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real d = w_prev[i] - w[i];
    f[i] = d;
    if (i >= dot_lo) acc += d * d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) p_out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(COSMO_BS) void k_aa_guard(int nparts, const real* __restrict__ p_in, real tau, AaFlags* __restrict__ F) {
  __shared__ real red[COSMO_BS / 64];
  const real na = sqrt(reduce_partials_sum(p_in, nparts, red));
  if (threadIdx.x == 0) {
    F->nrm_f_acc = na;
    F->declined = (na > F->nrm_f * tau) ? 1 : 0;
  }
}
// reset_accelerated_vector! (accelerator_interface.jl:129-134)
__global__ __launch_bounds__(COSMO_BS) void k_aa_reset(long long N, const real* __restrict__ g_last, real* __restrict__ w, real* __restrict__ w_prev) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) { const real g = g_last[i]; w[i] = g; w_prev[i] = g; }
}

// ---- normal-equations variants (Type1, Type2{NormalEquations}; RestartedMemory / RollingMemory) ------------------------------------------------
// update!: f = x - g ; first call after a restart: remember (x, g, f) ; otherwise X_j = x - x_last (Type1), G_j = g - g_last, F_j = f - f_last
__global__ __launch_bounds__(COSMO_BS) void k_aa_prep_ne(long long N, const real* __restrict__ g, const real* __restrict__ x, int init,
                                                         real* __restrict__ f, real* __restrict__ f_last, real* __restrict__ g_last, real* __restrict__ x_last,
                                                         real* __restrict__ Gj, real* __restrict__ Fj, real* __restrict__ Xj) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real gi = g[i], xi = x[i];
    const real fi = xi - gi;
    f[i] = fi;
    if (!init) {
      Gj[i] = gi - g_last[i];
      Fj[i] = fi - f_last[i];
      if (Xj) Xj[i] = xi - x_last[i];
    }
    g_last[i] = gi; f_last[i] = fi;
    if (x_last) x_last[i] = xi;
  }
}
// column j and row j of M = L' F over the first l columns: col[k] = <L_k, F_j>, row[k] = <L_j, F_k> (partials: slots k and AA_MAX_MEM + 1 + k)
template <int L>
__global__ __launch_bounds__(COSMO_BS) void k_aa_gram(long long N, int l, int j, const real* __restrict__ Lm, const real* __restrict__ Fm,
                                                      real* __restrict__ parts, long long dot_lo) {
  __shared__ real red[COSMO_BS / 64];
  real col[L], row[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { col[k] = 0.0; row[k] = 0.0; }
  const real* Fj = Fm + (size_t)j * N;
  const real* Lj = Lm + (size_t)j * N;
  for (long long i = dot_lo + (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real fj = Fj[i], lj = Lj[i];
#pragma unroll
    for (int k = 0; k < L; ++k) if (k < l) { col[k] += Lm[(size_t)k * N + i] * fj; row[k] += lj * Fm[(size_t)k * N + i]; }
  }
#pragma unroll
  for (int k = 0; k < L; ++k) {
    if (k < l) {
      const real a = block_sum(col[k], red);
      const real b = block_sum(row[k], red);
      if (threadIdx.x == 0) { parts[(size_t)k * COSMO_MAX_PARTIALS + blockIdx.x] = a; parts[(size_t)(AA_MAX_MEM + 1 + k) * COSMO_MAX_PARTIALS + blockIdx.x] = b; }
    }
  }
}
// one workgroup: M[:, j] and M[j, :] from the partials (column-major, leading dimension AA_MAX_MEM)
__global__ __launch_bounds__(COSMO_BS) void k_aa_gram_store(int l, int j, int nparts, const real* __restrict__ parts, AaFlags* __restrict__ F) {
  __shared__ real red[COSMO_BS / 64];
  for (int k = 0; k < l; ++k) {
    const real a = reduce_partials_sum(parts + (size_t)k * COSMO_MAX_PARTIALS, nparts, red);
    __syncthreads();
    const real b = reduce_partials_sum(parts + (size_t)(AA_MAX_MEM + 1 + k) * COSMO_MAX_PARTIALS, nparts, red);
    if (threadIdx.x == 0) { F->R[j * AA_MAX_MEM + k] = a; if (k != j) F->R[k * AA_MAX_MEM + j] = b; }
    __syncthreads();
  }
}
// one workgroup: eta = M[0..l, 0..l) \ (L' f) by LU with partial pivoting (LAPACK gesv on a copy); success unless a pivot is exactly zero / not finite
// (gesv's info > 0) or ||eta|| > eta_max or eta is not finite
__global__ __launch_bounds__(COSMO_BS) void k_aa_solve_ne(int l, int nparts, const real* __restrict__ parts, real eta_max, AaFlags* __restrict__ F) {
  __shared__ real red[COSMO_BS / 64];
  __shared__ real rhs[AA_MAX_MEM];
  __shared__ real Mw[AA_MAX_MEM * AA_MAX_MEM];
  for (int k = 0; k < l; ++k) {
    const real s = reduce_partials_sum(parts + (size_t)k * COSMO_MAX_PARTIALS, nparts, red);
    if (threadIdx.x == 0) rhs[k] = s;
    __syncthreads();
  }
  const real ff = reduce_partials_sum(parts + (size_t)AA_MAX_MEM * COSMO_MAX_PARTIALS, nparts, red);
  if (threadIdx.x == 0) {
    F->nrm_f = sqrt(ff);
    for (int c = 0; c < l; ++c) for (int r = 0; r < l; ++r) Mw[c * AA_MAX_MEM + r] = F->R[c * AA_MAX_MEM + r];
    bool ok = true;
    for (int c = 0; c < l && ok; ++c) {
      int pv = c; real best = fabs(Mw[c * AA_MAX_MEM + c]);
      for (int r = c + 1; r < l; ++r) { const real a = fabs(Mw[c * AA_MAX_MEM + r]); if (a > best) { best = a; pv = r; } }     // first maximum, as idamax
      const real piv = Mw[c * AA_MAX_MEM + pv];
      if (!(fabs(piv) <= REAL_MAX) || piv == R(0.0)) { ok = false; break; }
      if (pv != c) {
        for (int cc = 0; cc < l; ++cc) { const real t = Mw[cc * AA_MAX_MEM + c]; Mw[cc * AA_MAX_MEM + c] = Mw[cc * AA_MAX_MEM + pv]; Mw[cc * AA_MAX_MEM + pv] = t; }
        const real t = rhs[c]; rhs[c] = rhs[pv]; rhs[pv] = t;
      }
      for (int r = c + 1; r < l; ++r) {
        const real mlt = Mw[c * AA_MAX_MEM + r] / piv;
        Mw[c * AA_MAX_MEM + r] = mlt;
        for (int cc = c + 1; cc < l; ++cc) Mw[cc * AA_MAX_MEM + r] -= mlt * Mw[cc * AA_MAX_MEM + c];
        rhs[r] -= mlt * rhs[c];
      }
    }
    if (!ok) { F->success = 0; F->fail_singular += 1; return; }
    real nrm2 = 0.0;
    for (int i = l - 1; i >= 0; --i) {
      real s = rhs[i];
      for (int k = i + 1; k < l; ++k) s -= Mw[k * AA_MAX_MEM + i] * F->eta[k];
      const real e = s / Mw[i * AA_MAX_MEM + i];
      F->eta[i] = e;
      nrm2 += e * e;
    }
    const real en = sqrt(nrm2);
    F->eta_norm = en;
    if (!(en <= eta_max)) { F->success = 0; F->fail_eta += 1; return; }     // also catches NaN
    F->success = 1;
  }
}

inline AaState* aa_of(cosmo_hip_handle* h) { return static_cast<AaState*>(h->accel); }

}  // namespace

int32_t comm_allreduce_sum(cosmo_hip_handle* h, real* buf, size_t count);      // comm.hip
// row-sharded runs: the partial array(s) starting at `p` summed over the ranks, slot by slot (same count on every rank; entries beyond a
// rank's grid are zero from the allocation on)
static int32_t aa_share(cosmo_hip_handle* h, AaState* S, real* p) {
  if (!S->sharded) return COSMO_HIP_OK;
  return comm_allreduce_sum(h, p, (size_t)S->grid);
}

void aa_free(cosmo_hip_handle* h) {
  AaState* S = aa_of(h);
  if (!S) return;
  for (real* p : {S->G, S->Q, S->f, S->f_last, S->g_last, S->parts, S->X, S->x_last}) if (p) (void)hipFree(p);
  if (S->flags) (void)hipFree(S->flags);
  if (S->flags_host) (void)hipHostFree(S->flags_host);
  delete S;
  h->accel = nullptr;
}

bool aa_enabled(const cosmo_hip_handle* h) { return h->accel != nullptr; }
bool aa_get_params(const cosmo_hip_handle* h, cosmo_hip_accel_params* out) { const AaState* S = static_cast<const AaState*>(h->accel); if (!S) return false; *out = S->prm; return true; }

int32_t aa_restart(cosmo_hip_handle* h) {      // CA.restart! -> empty_history!
  AaState* S = aa_of(h);
  if (!S) return COSMO_HIP_OK;
  const size_t slab = sizeof(real) * (size_t)S->N * (size_t)S->mem;
  HIPCHK(h, hipMemsetAsync(S->G, 0, slab, h->stream));
  HIPCHK(h, hipMemsetAsync(S->Q, 0, slab, h->stream));
  HIPCHK(h, hipMemsetAsync(S->f, 0, sizeof(real) * (size_t)S->N, h->stream));
  HIPCHK(h, hipMemsetAsync(S->f_last, 0, sizeof(real) * (size_t)S->N, h->stream));
  HIPCHK(h, hipMemsetAsync(S->g_last, 0, sizeof(real) * (size_t)S->N, h->stream));
  if (S->X) HIPCHK(h, hipMemsetAsync(S->X, 0, slab, h->stream));
  if (S->x_last) HIPCHK(h, hipMemsetAsync(S->x_last, 0, sizeof(real) * (size_t)S->N, h->stream));
  HIPCHK(h, hipMemsetAsync(S->flags, 0, sizeof(AaFlags), h->stream));
  S->iter = 0;
  S->init_phase = true;
  return COSMO_HIP_OK;
}

extern "C" void cosmo_hip_default_accel_params(cosmo_hip_accel_params* p) {
  if (!p) return;
  p->kind = COSMO_HIP_ACCEL_ANDERSON; p->mem = 15; p->min_mem = 3; p->safeguard = 1;
  p->start_iter = 2; p->safeguard_tol = 2.0; p->eta_max = 1e4; p->start_accuracy = -1.0;
}

extern "C" int32_t cosmo_hip_set_accelerator(cosmo_hip_handle* h, const cosmo_hip_accel_params* p) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_accelerator: set_problem first");
  aa_free(h);
  if (!p || p->kind == COSMO_HIP_ACCEL_EMPTY) return COSMO_HIP_OK;
  if (p->kind < COSMO_HIP_ACCEL_ANDERSON || p->kind > COSMO_HIP_ACCEL_ANDERSON_TYPE2NE_ROLLING) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "accelerator kind %d", (int)p->kind);
  if (p->mem < 1 || p->mem > AA_MAX_MEM || p->min_mem < 1 || p->start_iter < 2 || !(p->safeguard_tol >= 0.0) || !(p->eta_max > 0.0))
    return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_accelerator: need 1 <= mem <= %d, min_mem >= 1, start_iter >= 2", AA_MAX_MEM);
  AaState* S = new AaState();
  h->accel = S;
  S->prm = *p;
  S->ne = p->kind != COSMO_HIP_ACCEL_ANDERSON;
  S->type1 = p->kind == COSMO_HIP_ACCEL_ANDERSON_TYPE1_RESTARTED || p->kind == COSMO_HIP_ACCEL_ANDERSON_TYPE1_ROLLING;
  S->rolling = p->kind == COSMO_HIP_ACCEL_ANDERSON_TYPE1_ROLLING || p->kind == COSMO_HIP_ACCEL_ANDERSON_TYPE2NE_ROLLING;
  S->N = h->n + h->m;                           // row-sharded handle: h->m is the LOCAL row count, w = [x ; this rank's rows]
  S->sharded = h->row_shard && comm_nranks(h) > 1;
  S->dot_lo = (S->sharded && comm_rank(h) > 0) ? h->n : 0;
  // mem = min(mem, dim) with dim = n + m of the WHOLE problem: on a row-sharded handle the local length differs per rank, and a rank with a
  // different mem would restart on a different schedule and issue a different number of all-reduces than its peers (a hang inside RCCL)
  const long long dim_g = S->sharded ? h->n + h->m_g : S->N;
  S->mem = (int)std::min<long long>(p->mem, std::max<long long>(dim_g, 1));
  const size_t N = (size_t)std::max<long long>(S->N, 1);
  HIPCHK(h, hipMalloc((void**)&S->G, sizeof(real) * N * S->mem));
  HIPCHK(h, hipMalloc((void**)&S->Q, sizeof(real) * N * S->mem));
  HIPCHK(h, hipMalloc((void**)&S->f, sizeof(real) * N));
  HIPCHK(h, hipMalloc((void**)&S->f_last, sizeof(real) * N));
  HIPCHK(h, hipMalloc((void**)&S->g_last, sizeof(real) * N));
  if (S->type1) {
    HIPCHK(h, hipMalloc((void**)&S->X, sizeof(real) * N * S->mem));
    HIPCHK(h, hipMalloc((void**)&S->x_last, sizeof(real) * N));
  }
  HIPCHK(h, hipMalloc((void**)&S->parts, sizeof(real) * (2 * AA_MAX_MEM + 2) * COSMO_MAX_PARTIALS));
  HIPCHK(h, hipMemsetAsync(S->parts, 0, sizeof(real) * (2 * AA_MAX_MEM + 2) * COSMO_MAX_PARTIALS, h->stream));   // slots beyond `grid` stay zero (all-reduced whole)
  HIPCHK(h, hipMalloc((void**)&S->flags, sizeof(AaFlags)));
  HIPCHK(h, hipHostMalloc((void**)&S->flags_host, sizeof(AaFlags)));
  memset(S->flags_host, 0, sizeof(AaFlags));
  long long g = (S->N + (long long)COSMO_BS * 4 - 1) / ((long long)COSMO_BS * 4);
  S->grid = (int)std::max<long long>(1, std::min<long long>(g, 1024));
  // row-sharded: the SAME number of partials on every rank (the local lengths differ): a shorter rank's surplus workgroups contribute +0.0, and
  // no slot of the all-reduced array ever keeps a stale sum of an earlier, longer reduction
  if (S->sharded) S->grid = 1024;
  S->nparts = S->grid;
  CHK(aa_restart(h));
  S->active = false;
  return COSMO_HIP_OK;
}

// optimize! entry: a second optimize! restarts the accelerator and deactivates it (src/setup.jl:47-49)
int32_t aa_begin_solve(cosmo_hip_handle* h) {
  AaState* S = aa_of(h);
  if (!S) return COSMO_HIP_OK;
  CHK(aa_restart(h));
  S->active = false;
  S->num_accelerated = S->num_restarts = S->num_declined = S->num_accepted = 0;
  S->num_rho_restarts = 0;
  return COSMO_HIP_OK;
}

// acceleration_pre! (accelerator_interface.jl:58-76): activation, CA.update!(w, w_prev), CA.accelerate!(w).
// *attempted = an accelerate! kernel chain was enqueued and the success flag has to be fetched after the next sync.
int32_t aa_enqueue_pre(cosmo_hip_handle* h, long long it, bool* attempted) {
  AaState* S = aa_of(h);
  *attempted = false;
  if (!S) return COSMO_HIP_OK;
  if (!S->active && !(S->prm.start_accuracy >= 0.0) && it >= S->prm.start_iter) S->active = true;   // check_activation! (Immediate / IterActivation)
  if (!S->active) return COSMO_HIP_OK;
  const long long N = S->N;
  const dim3 G(S->grid), B(COSMO_BS);
  hipStream_t st = h->stream;
  // ---- update! ----
  if (S->ne) {
    // Type1 / Type2{NormalEquations}: the histories X (Type1), F (in the Q slab), G; M = L' F kept current (L = X or F)
    const real* Lm = S->type1 ? S->X : S->Q;
    if (S->init_phase) {
      hipLaunchKernelGGL(k_aa_prep_ne, G, B, 0, st, N, h->w, h->w_prev, 1, S->f, S->f_last, S->g_last, S->x_last, S->G, S->Q, S->X);
      S->init_phase = false;
    } else {
      int j = S->iter % S->mem;
      if (j == 0 && S->iter != 0 && !S->rolling) {       // RestartedMemory: the memory is full -> start over (RollingMemory: column j, the oldest, is overwritten)
        const size_t slab = sizeof(real) * (size_t)N * (size_t)S->mem;
        HIPCHK(h, hipMemsetAsync(S->G, 0, slab, st));
        HIPCHK(h, hipMemsetAsync(S->Q, 0, slab, st));
        if (S->X) HIPCHK(h, hipMemsetAsync(S->X, 0, slab, st));
        HIPCHK(h, hipMemsetAsync(S->flags, 0, offsetof(AaFlags, nrm_f), st));
        HIPCHK(h, hipMemsetAsync(reinterpret_cast<char*>(S->flags) + offsetof(AaFlags, R), 0, sizeof(real) * AA_MAX_MEM * AA_MAX_MEM, st));
        S->iter = 0;
        S->num_restarts += 1;
      }
      hipLaunchKernelGGL(k_aa_prep_ne, G, B, 0, st, N, h->w, h->w_prev, 0, S->f, S->f_last, S->g_last, S->x_last, S->G + (size_t)j * N, S->Q + (size_t)j * N,
                         S->X ? S->X + (size_t)j * N : (real*)nullptr);
      S->iter += 1;
      const int ln = std::min(S->iter, S->mem);           // columns in the history, the new one included
      if (ln <= 8) hipLaunchKernelGGL((k_aa_gram<8>), G, B, 0, st, N, ln, j, Lm, S->Q, S->parts, S->dot_lo);
      else if (ln <= 16) hipLaunchKernelGGL((k_aa_gram<16>), G, B, 0, st, N, ln, j, Lm, S->Q, S->parts, S->dot_lo);
      else hipLaunchKernelGGL((k_aa_gram<AA_MAX_MEM>), G, B, 0, st, N, ln, j, Lm, S->Q, S->parts, S->dot_lo);
      if (S->sharded) {
        CHK(comm_allreduce_sum(h, S->parts, (size_t)ln * COSMO_MAX_PARTIALS));
        CHK(comm_allreduce_sum(h, AA_PARTS(S, AA_MAX_MEM + 1), (size_t)ln * COSMO_MAX_PARTIALS));
      }
      hipLaunchKernelGGL(k_aa_gram_store, dim3(1), B, 0, st, ln, j, S->nparts, S->parts, S->flags);
    }
  } else
  if (S->init_phase) {
    hipLaunchKernelGGL(k_aa_prep, G, B, 0, st, N, h->w, h->w_prev, 1, S->f, S->f_last, S->g_last, S->G, S->Q, S->Q, 0, AA_PARTS(S, 0), S->dot_lo);
    S->init_phase = false;
  } else {
    int j = S->iter % S->mem;
    if (j == 0 && S->iter != 0) {       // RestartedMemory: the memory is full -> start over
      const size_t slab = sizeof(real) * (size_t)N * (size_t)S->mem;
      HIPCHK(h, hipMemsetAsync(S->G, 0, slab, st));
      HIPCHK(h, hipMemsetAsync(S->Q, 0, slab, st));
      HIPCHK(h, hipMemsetAsync(S->flags, 0, offsetof(AaFlags, nrm_f), st));
      HIPCHK(h, hipMemsetAsync(reinterpret_cast<char*>(S->flags) + offsetof(AaFlags, R), 0, sizeof(real) * AA_MAX_MEM * AA_MAX_MEM, st));
      S->iter = 0;
      S->num_restarts += 1;
      j = 0;
    }
    real* Gj = S->G + (size_t)j * N;
    real* v = S->Q + (size_t)j * N;
    real* Rcol = reinterpret_cast<real*>(reinterpret_cast<char*>(S->flags) + offsetof(AaFlags, R)) + (size_t)j * AA_MAX_MEM;
    hipLaunchKernelGGL(k_aa_prep, G, B, 0, st, N, h->w, h->w_prev, 0, S->f, S->f_last, S->g_last, Gj, v, S->Q, j, AA_PARTS(S, 0), S->dot_lo);
    CHK(aa_share(h, S, AA_PARTS(S, 0)));
    for (int i = 0; i < j; ++i) {
      const int last = (i + 1 == j);
      hipLaunchKernelGGL(k_aa_mgs, G, B, 0, st, N, S->nparts, AA_PARTS(S, i & 1), S->Q + (size_t)i * N, S->Q + (size_t)(last ? i : i + 1) * N, v, last,
                         Rcol + i, AA_PARTS(S, (i + 1) & 1), S->dot_lo);
      CHK(aa_share(h, S, AA_PARTS(S, (i + 1) & 1)));
    }
    hipLaunchKernelGGL(k_aa_normalize, G, B, 0, st, N, S->nparts, AA_PARTS(S, j & 1), v, Rcol + j);
    S->iter += 1;
  }
  // ---- accelerate! ----
  const int l = std::min(S->iter, S->mem);
  if (l < S->prm.min_mem) {
    HIPCHK(h, hipMemsetAsync(&S->flags->success, 0, sizeof(int), st));
    HIPCHK(h, hipGetLastError());
    return COSMO_HIP_OK;
  }
  const real* Lq = (S->ne && S->type1) ? S->X : S->Q;     // Q' f (default), X' f (Type1), F' f (Type2{NormalEquations}: F lives in the Q slab)
  if (l <= 8) hipLaunchKernelGGL((k_aa_qtf<8>), G, B, 0, st, N, l, Lq, S->f, S->parts, S->dot_lo);
  else if (l <= 16) hipLaunchKernelGGL((k_aa_qtf<16>), G, B, 0, st, N, l, Lq, S->f, S->parts, S->dot_lo);
  else hipLaunchKernelGGL((k_aa_qtf<AA_MAX_MEM>), G, B, 0, st, N, l, Lq, S->f, S->parts, S->dot_lo);
  if (S->sharded) {                                     // Q'f (the l written slots, contiguous) and ||f||^2 (slot AA_MAX_MEM)
    CHK(comm_allreduce_sum(h, S->parts, (size_t)l * COSMO_MAX_PARTIALS));
    CHK(comm_allreduce_sum(h, AA_PARTS(S, AA_MAX_MEM), (size_t)S->grid));
  }
  if (S->ne) hipLaunchKernelGGL(k_aa_solve_ne, dim3(1), B, 0, st, l, S->nparts, S->parts, S->prm.eta_max, S->flags);
  else hipLaunchKernelGGL(k_aa_solve, dim3(1), B, 0, st, l, S->nparts, S->parts, S->prm.eta_max, S->flags);
  if (l <= 8) hipLaunchKernelGGL((k_aa_apply<8>), G, B, 0, st, N, l, S->G, S->flags, h->w);
  else if (l <= 16) hipLaunchKernelGGL((k_aa_apply<16>), G, B, 0, st, N, l, S->G, S->flags, h->w);
  else hipLaunchKernelGGL((k_aa_apply<AA_MAX_MEM>), G, B, 0, st, N, l, S->G, S->flags, h->w);
  HIPCHK(h, hipGetLastError());
  *attempted = true;
  return COSMO_HIP_OK;
}

// after a stream synchronisation
int32_t aa_fetch_flags(cosmo_hip_handle* h, int* success, int* declined) {
  AaState* S = aa_of(h);
  if (!S) { if (success) *success = 0; if (declined) *declined = 0; return COSMO_HIP_OK; }
  HIPCHK(h, hipMemcpyAsync(S->flags_host, S->flags, offsetof(AaFlags, R), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (success) *success = S->flags_host->success;
  if (declined) *declined = S->flags_host->declined;
  return COSMO_HIP_OK;
}

// acceleration_post! part 1 (accelerator_interface.jl:85-100): residual of the accelerated point against tau * ||f||
int32_t aa_enqueue_guard(cosmo_hip_handle* h) {
  AaState* S = aa_of(h);
  hipLaunchKernelGGL(k_aa_resnorm, dim3(S->grid), dim3(COSMO_BS), 0, h->stream, S->N, h->w, h->w_prev, S->f, AA_PARTS(S, 0), S->dot_lo);
  CHK(aa_share(h, S, AA_PARTS(S, 0)));
  hipLaunchKernelGGL(k_aa_guard, dim3(1), dim3(COSMO_BS), 0, h->stream, S->nparts, AA_PARTS(S, 0), S->prm.safeguard_tol, S->flags);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}
int32_t aa_enqueue_reset(cosmo_hip_handle* h) {
  AaState* S = aa_of(h);
  hipLaunchKernelGGL(k_aa_reset, dim3(S->grid), dim3(COSMO_BS), 0, h->stream, S->N, S->g_last, h->w, h->w_prev);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}
bool aa_safeguarded(const cosmo_hip_handle* h) { const AaState* S = static_cast<const AaState*>(h->accel); return S && S->prm.safeguard != 0; }
// check_activation!(ws, ::AccuracyActivation, r::ResultInfo) (accelerator_interface.jl:38-46), called from has_converged at
// every termination check
void aa_check_accuracy_activation(cosmo_hip_handle* h, real r_prim, real r_dual, real max_norm_prim, real max_norm_dual) {
  AaState* S = aa_of(h);
  if (!S || S->active || !(S->prm.start_accuracy >= 0.0)) return;
  const real tol = S->prm.start_accuracy;
  if (r_prim < tol + tol * max_norm_prim && r_dual < tol + tol * max_norm_dual) S->active = true;
}
bool aa_active(const cosmo_hip_handle* h) { const AaState* S = static_cast<const AaState*>(h->accel); return S && S->active; }
void aa_count(cosmo_hip_handle* h, int accelerated, int declined) {
  AaState* S = aa_of(h);
  if (!S) return;
  S->num_accelerated += accelerated;
  if (accelerated && S->prm.safeguard) { if (declined) S->num_declined += 1; else S->num_accepted += 1; }
}

void aa_note_rho_restart(cosmo_hip_handle* h) { AaState* S = aa_of(h); if (S) S->num_rho_restarts += 1; }
// out = {restarts because the memory was full (RestartedMemory), restarts because rho was adapted} of the last optimize -- the second is what the reference's
// test/UnitTests/AccelerationTests/adaptive_rho_acc_restarts.jl counts in the accelerator's log (`:rho_adapted` entries == num_rho_adaptions)
extern "C" int32_t cosmo_hip_get_accel_restarts(cosmo_hip_handle* h, int64_t out[2]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  AaState* S = aa_of(h);
  out[0] = S ? S->num_restarts : 0; out[1] = S ? S->num_rho_restarts : 0;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_get_accel_stats(cosmo_hip_handle* h, int64_t out[6]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  AaState* S = aa_of(h);
  for (int i = 0; i < 6; ++i) out[i] = 0;
  out[5] = h->safeguarding_iter;
  if (!S) return COSMO_HIP_OK;
  out[0] = S->num_accelerated; out[1] = S->num_accepted; out[2] = S->num_declined; out[3] = S->num_restarts; out[4] = S->active ? 1 : 0;
  return COSMO_HIP_OK;
}
