// psd_polar.hip -- PSD projection of LARGE cones (d > 256, one cone at a time) and of MID-SIZE cones (64 < d <= 256, batched) as a
// matrix-sign (polar) iteration made only of symmetric matrix products on the fp64 matrix cores.
//
// Reference semantics (src/convexset.jl:219-263): X+ = sum_{lambda_j > 0} lambda_j z_j z_j'.  For symmetric X this is
//     X+ = (X + |X|) / 2 ,  |X| = sign(X) X ,  sign(X) = Z sign(Lambda) Z'
// so the projection needs the matrix sign U = sign(X), not the eigenvectors.  U is the limit of odd quintic fixed-point steps
//     U <- U (a I + b U^2 + c U^4),  U_0 = 2 X / ||X||_F   (all iterates are polynomials in X: symmetric, commuting, spectrum in [0, 2])
// with PER-STEP minimax-optimal coefficients (round 2; design tool tools/polar_schedule.py, which the CPU test
// tests/test_polar_schedule.py replays against the table exported by cosmo_hip_polar_schedule):
//   lifting   (k_lift steps, default 10): the minimax odd quintic p for [l*, 2.1], l* = 0.02212 chosen so that p maps [l*, 2.1] into
//             [0.085, 1.915] (5 % margin at the top: rounding cannot push an eigenvalue out of the domain; the interior minimum
//             of p equals p(l*) = 0.085, so an eigenvalue that is already lifted can never fall back below the finishing range --
//             the l -> 0 limit polynomial, slope 4.05, has an interior ROOT and is unusable); slope 3.8438 at 0 and
//             p(x) >= 0.9996 * 3.8438 x on [0, l*], so k steps lift every |lambda| >= 0.0425 * 3.8426^-k ||X||_F above 0.085
//             (k = 10: 6e-8 ||X||_F; the round-1 fixed triple had slope 3.44 and needed 20 steps for its 3e-12);
//   finishing (5 steps): the greedy minimax sequence for [0.08, 2.02] -> [0.28, 1.72] -> [0.71, 1.29] -> 1 +- 1.6e-2 -> 1 +- 1e-5
//             -> 1 +- 2e-15 (the last two are the Newton-Schulz quintic (15, -10, 3) / 8, cubically convergent).
// A-posteriori VERIFICATION (one extra product): with H = U X (needed anyway) the matrix G = U H - X = (U^2 - I) X has
//     ||G||_F^2 = sum lambda_i^2 (1 - u_i^2)^2  >=  sum lambda_i^2 (1 - |u_i|)^2 = 4 ||X+ - X+_exact||_F^2,
// so ||G||_F / 2 bounds the projection error rigorously (up to the rounding of the products, ~1e-15 ||X||_F).  If it exceeds
// 8 d eps ||X||_F (1/8 of the parity tolerance 64 d eps), FALLBACK rounds (3 more lifting steps + the finishing steps, verified
// again) run.  Two ways to schedule them (PolarPlan::speculate): ON DEMAND (default for a plan with at most two large cones, and for
// the batch) -- after the main schedule the host reads the verification flag, ONE small stream synchronisation per large cone / per
// batch and projection, and enqueues a round only if it is needed, so the loop is NOT sync-free while such cones are present; or
// SPECULATIVELY (default from three large cones on, COSMO_HIP_POLAR_SPECULATE=1 forces it) -- both rounds are always enqueued behind
// a device-side gate, 26 no-op launches per cone and round, and a projection involves no host synchronisation at all (a problem with
// many d > 256 cones would otherwise drain the pipeline once per cone in every iteration).  Eigenvalues so small that they never lift (|lambda| below
// ~1e-12 ||X||_F) perturb X+ by less than their own magnitude and pass the verification by construction.
// rank = round((tr U + tr U^2) / 2) (exact when every eigenvalue was lifted; exact zeros count as not positive, like lambda > 0).
//
// Why this shape on MI355X: one-sided Jacobi at d = 2000 is ~3000 dependent tournament rounds of latency-bound 16x16
// rotations (135-180 ms, 1 % MFMA-busy, profiles/r01_psd_mfma_counters.json); a tridiagonal QL/D&C chain is serial.  The sign
// iteration is (k_lift + 5) * 3 + 2 = 47 products of d x d symmetric matrices whose result is symmetric, so only the upper
// tiles are computed (d^3 flops per product) and mirrored: fixed main schedule, every kernel guarded by ctl->halt like the rest of
// the loop; host involvement only as described above for the fallback decision.
//
// Kernel: k_symm_gemm -- C = alpha A B + beta Cin on the upper tiles, A and B symmetric (so both operands are read
// "contiguous along the output index, strided along k"), 64x64 tile per workgroup of 4 waves (32x32 per wave = 2x2
// v_mfma_f64_16x16x4_f64 accumulators), k-panels of 16 staged through double-buffered LDS with an 80-real row pitch
// (conflict-free ds_read_b64 for the 16-lane x 4-row operand fragments), global loads of panel k+1 in flight during
// the MFMAs of panel k.
#include "psd_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>

#define PK 16   // k panel
#ifndef POLAR_PREFETCH_DEPTH
#define POLAR_PREFETCH_DEPTH 2   // operand panels in flight per workgroup in the tile main loop (2 | 3).  3 was measured: no gain at d = 2000
                                 // (167.8 vs 167.4 us per product), slower in the batched kernel (51.1 vs 47.5 us): profiles/r02_prefetch_depth.txt
#endif
#define PLD 80  // LDS row pitch in doubles

struct PolarCone {
  int idx;            // index in PsdPlan::cones
  int off, d, kind;
  int ts;             // tile side of the products (64 or 96), chosen per cone
  int sk;             // 2: intra-workgroup split of k (8 waves), when the upper tiles give at most one tile per CU; 3: stream-K
  int skg, skc;       // stream-K: workgroups of a launch, ticket classes (8 = one tile range per XCD, 1 = a single range)
  int ld;             // d rounded up to ts
  long long woff;     // offset of this cone's 4 work matrices (doubles)
};

struct BatchCone { long long woff; int off, d, kind, ld, idx, ts; };   // a mid-size cone of the batched path (device table); ts = tile side of its products (64 | 96)

// device-side control of the verification / fallback rounds (one per plan; the large cones are projected one after the other)
// stream-K product (k_symm_gemm_sk): ticket counter, spin-timeout marker, flag[G] = epoch of the slot's last partial tile
struct SkSync { unsigned ticket; unsigned timeout; unsigned pad[14]; unsigned flag[1]; };
struct PolarDev {
  int gate;          // 1: the last verification failed, the next guarded round must run
  int rounds;        // fallback rounds executed so far
  int verified;      // projections (large cone or whole batch) whose error bound passed
  int unverified;    // ... that still failed after the last enqueued round
  int projections;
  int pad[3];
  real err_last, err_max;   // ||G||_F / (2 ||X||_F) of the last verification, max over all
};

// schedule constants (tools/polar_schedule.py)
static const real kPolarLift[3] = {3.8438259784376458, -2.5414431444903771, 0.42553478492344637};   // minimax on [0.02212, 2.1]: range [0.085, 1.915]
#define POLAR_NFIN 5
static const real kPolarFinish[POLAR_NFIN][3] = {
    {3.4931704678738202, -2.370790022242379, 0.42240833954524776},    // [0.08, 2.02]   -> 1 +- 0.722
    {2.7077959763108326, -2.015093165601975, 0.45683257965863566},    // [0.278, 1.722] -> 1 +- 0.289
    {1.9693651031668091, -1.3512768962961363, 0.3852630352029357},    // [0.711, 1.289] -> 1 +- 1.56e-2
    {1.875, -1.25, 0.375},                                            // Newton-Schulz   -> 1 +- 9.5e-6
    {1.875, -1.25, 0.375}};                                           //                 -> 1 +- 2.1e-15
#define POLAR_RLIFT 3        // lifting steps of a fallback round (RLIFT + NFIN is even: the buffer parity is preserved)
// Float32, large cones (d > 256): the last Newton-Schulz step (1 +- 9.5e-6 -> 1 +- 2e-15) is below the resolution of the element type and
// of the verification threshold 8 d eps32 >= 1.2e-4, so the schedule stops after four finishing steps; the fallback rounds lift four
// times (4 + 4 steps: even, like 3 + 5).  The batched path (d down to 17: threshold 8e-6) keeps all five.
#define POLAR_NFIN_LARGE (REAL_IS_FLOAT ? 4 : POLAR_NFIN)
#define POLAR_RLIFT_LARGE (REAL_IS_FLOAT ? 4 : POLAR_RLIFT)

struct PolarPlan {
  std::vector<PolarCone> cones;
  real* W = nullptr;       // 4 * ld^2 doubles per cone: X, U, Y, T
  real* parts = nullptr;   // per cone COSMO_MAX_PARTIALS norm partials + trace partials
  real* nrm = nullptr;     // per cone ||X||_F
  // lifting steps of the main schedule (COSMO_HIP_POLAR_KLIFT; grows by 3 when fallbacks are frequent).  Float64: 9 (every
  // |lambda| >= 2.3e-7 ||X||_F lifted; 10 until round 3 -- the replay of real spectra, tests/studies/lift_depth_replay.py, finds the minimal passing
  // depth <= 9 in every one of the first 100 projections of BASELINE config 4 and in 66 of 70 of config 5 (4 repair trains instead of 1); late
  // iterates of a convergent run need 11-12, where the grow-by-3 rule of polar_adapt takes over as it did from 10).  Float32: 5 -- the verification threshold 8 d eps scales with eps(Float32) (2e-4 ... 2e-3 for
  // d = 200 ... 2000), eigenvalues below it pass by construction, and 0.0425 * 3.84^-5 = 5e-5 is already beneath it
  int k_lift = REAL_IS_FLOAT ? 5 : 9;
  int max_rounds = 2;        // guarded fallback rounds enqueued per projection
  int rescale = 1;           // spectral rescaling in the first step of a large cone's iteration (COSMO_HIP_POLAR_RESCALE=0 disables)
  int batch_occ = 3;         // register-allocation variant of the batched product kernels: 3 or 4 waves per SIMD (COSMO_HIP_POLAR_BATCH_OCC; the plan sets 4 for the ragged kernel)
  real tol_factor = 8.0;   // verification threshold tol_factor * d * eps (relative to ||X||_F)
  PolarDev* dev = nullptr;
  PolarDev seen;             // host copy at the last polar_adapt
  long long launches[4] = {0, 0, 0, 0};   // <64,1>, <96,1>, <96,2> or stream-K, batch
  // stream-K product of the large cones (k_symm_gemm_sk): scratch slots for the partial tiles, ticket + flags
  int streamk = 0;           // COSMO_HIP_POLAR_STREAMK=1: stream-K product kernel (k_symm_gemm_sk) instead of one tile per workgroup; measured
                             // slower (see the kernel's header), so opt-in
  real* sk_scratch = nullptr;
  SkSync* sk_sync = nullptr;
  int sk_gmax = 0;
  unsigned sk_base = 0, sk_epoch = 0;   // first ticket of the next launch (mod 2^32), launch counter
  int products_last_large = 0, products_last_batch = 0;
  // batch of mid-size cones (one launch per product for all of them)
  std::vector<BatchCone> bcones;
  BatchCone* d_bcones = nullptr;
  int4* d_btiles = nullptr;   // tile descriptors of the cones with 64 x 64 tiles, then (from nbtiles on) of the cones with 96 x 96 tiles
  int nbtiles = 0, nbtiles96 = 0;
  int speculate = 0;         // 0: after the main schedule the host reads the verification flag (one small synchronisation per projection) and enqueues a
                             // fallback round only if it is needed; COSMO_HIP_POLAR_SPECULATE=1: both rounds are always enqueued behind device-side gates
                             // (no host synchronisation inside a projection, 52 no-op launches).  Measured: cfg4 127.9 -> 129.0, cfg5 155.1 -> 157.7 it/s
                             // without speculation (profiles/r02_fallback_speculation.txt)
  int* gate_host = nullptr;  // pinned copy of PolarDev::gate for the non-speculative mode
  int batch_wave = 0;        // COSMO_HIP_POLAR_BATCH_WAVE=1: wave-per-tile product kernel (k_symm_gemm_batch_w) for the 64 x 64 tile class.  Bit-identical
                             // to the workgroup-per-tile kernel; measured on BASELINE config 5: 50.7 vs 47.3 us per product, 150.2 vs 154.7 it/s => opt-in
  int batch_ragged = 1;      // block-balanced ragged tiles (k_symm_gemm_batch_r); COSMO_HIP_POLAR_BATCH_RAGGED=0: the 64 x 64 quadrant kernel
  void* d_rtiles = nullptr;  // RTile list of the ragged kernel (XCD-interleaved like d_btiles)
  int nrtiles = 0;
  // persistent dependency-driven main schedule (k_polar_dataflow; COSMO_HIP_POLAR_DATAFLOW): the XCDs' tile lists one after the other, per-cone tile counts,
  // ticket / completion counters
  int dataflow = 0;
  void* d_df_tiles = nullptr;
  int* d_df_cone_nt = nullptr;
  unsigned* d_df_sync = nullptr;
  int df_xoff[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int df_grid = 0;
  long long df_launches = 0, df_timed = 0;
  int df_nprod = 0;
  double df_seconds = 0.0;   // event-timed launches (df_timed of them)
  hipEvent_t df_ev[2] = {nullptr, nullptr};
  int df_ev_pending = 0;
  // COMPACT REPAIR (round 5): a failed verification is repaired by launches over the FAILING cones' tiles only (tile list rebuilt on the host from the
  // per-cone verification flags it has just read, uploaded from a pinned buffer), not by 26 full-grid launches in which all other tiles look at a
  // gate and leave; the first repair round resumes with ONE lifting step (a cone that fails at its adaptive depth is one step short in the replay,
  // tests/studies/lift_depth_replay.py), the second with the POLAR_RLIFT of before.  Used by the adaptive per-cone depth mode.
  void* h_rtiles = nullptr;  // host copy of the RTile list (malloc)
  void* rep_host = nullptr;  // pinned staging of the compact list
  void* d_rtiles_rep = nullptr;
  int nrep = 0;
  long long repair_launch_tiles = 0, repair_trains = 0;
  double batch_flops_performed = 0.0, batch_flops_useful = 0.0;   // per product: 16 x 16 x 16 blocks actually issued; sum d^2 (d + 1)
  int batch_ts96 = 0;        // COSMO_HIP_POLAR_BATCH_TS96=1: cones whose side fits 96 / 192 take 96 x 96 tiles in a second launch per product.
                             // Measured on BASELINE config 5: SLOWER, 62.2 vs 47.0 us per product, 133.9 vs 155.2 it/s (two launch tails per
                             // product, two workgroups per CU for the 96-class) => opt-in; profiles/r02_batch_tile_classes.txt
  real* BW = nullptr;
  real* bparts = nullptr;
  real* bnrm = nullptr;    // per batched cone ||X||_F
  int* bgate = nullptr;      // per batched cone: 1 = its verification failed, the next fallback round processes it
  // ---- per-cone lifting depth of the batch (round 4) ------------------------------------------------------------------------------------
  // Every cone of the batch runs ITS OWN number of lifting steps bk[c] (cone c joins the batch's schedule at step kmax - bk[c]: the gate table
  // lgate[t * n + c] = (t >= kmax - bk[c]) is handed to the product kernels as the per-cone gate they already have for the fallback rounds) and
  // the depth follows the cone's own verification history: bk[c] - 1 after `bm[c]` consecutive verified projections, bk[c] + 1 (and the cone
  // alone takes a fallback round, as before) on a failed verification, after which the cone waits twice as long before it probes downwards
  // again.  The a-posteriori check is unchanged, so the error bound of every projection is the same rigorous 8 d eps ||X||_F.  Replay on the
  // spectra of BASELINE config 5 (tests/studies/lift_depth_replay.py, profiles/r04_lift_depth_replay.txt): minimal passing depth 3-5 for
  // 90 % of the (iteration, cone) pairs against the fixed 10.  A function of each cone's own history only => independent of how the cones are
  // partitioned over ranks.
  // MEASURED (profiles/r04_adaptive_lifting_depth.txt) and therefore OPT-IN (COSMO_HIP_POLAR_ADAPT=1; default: fixed depth k_lift for all cones,
  // the schedule of round 3): on BASELINE config 5 the d^3-weighted products per projection fall from 47 to 36.6, yet the step gets SLOWER (223.9 vs
  // 230.6 it/s): a cone's minimal passing depth moves by +-1 from one ADMM iterate to the next (18 % of the steps go up), so with 400 cones some
  // cone fails its verification in nearly every projection (39 of 50), and every such projection pays the repair train -- 26 gated launches and a
  // second host synchronisation -- whatever the size of the failing cones.  The replay with any history-based rule (running maximum of the last
  // 10 depths + 1: 58 of 60 projections still contain a failure) says the same.  Config 4 (one cone): depth 9 instead of 10 after 50 projections,
  // 128.8 vs 129.0 it/s.
  int adapt = 0;
  int compact_repair = 1;                // COSMO_HIP_POLAR_COMPACT_REPAIR=0: the full-grid gated repair train of round 4
  std::vector<int> bk, bstreak, bm;     // per batched cone: lifting depth, verified projections since the last change, patience
  int* d_lgate = nullptr;                // (LIFT_CAP x n) gate table of the lifting steps
  int* d_ubuf = nullptr;                 // per cone: work buffer (1 | 2) that receives U_0 (the one that is `iu` when the cone joins)
  int* bgate_host = nullptr;             // pinned: per-cone verification result of round 0
  int lgate_kmax = -1;                   // kmax the device tables were built for (-1: dirty)
  long long depth_fail = 0, depth_down = 0, depth_proj = 0;
  int seen_proj_batch = 0;               // PolarDev::projections at the last depth-control step
  std::vector<int> lk, lstreak, lm;      // the same control per LARGE cone (one cone per projection: the depth is simply its main schedule)
  int seen_proj_large = 0;
  double wprod_last = 0.0;               // sum_c d_c^3 * products of cone c / sum_c d_c^3 of the last projection (main schedule + verification)
};
#define POLAR_LIFT_CAP 18

extern "C" int32_t cosmo_hip_polar_schedule(int32_t k_lift, double* abc, int32_t* nsteps) {
  if (k_lift < 0 || k_lift > 64 || !nsteps) return COSMO_HIP_ERR_INVALID;
  *nsteps = k_lift + POLAR_NFIN;
  if (!abc) return COSMO_HIP_OK;
  for (int t = 0; t < k_lift; ++t) for (int q = 0; q < 3; ++q) abc[3 * t + q] = (double)kPolarLift[q];
  for (int t = 0; t < POLAR_NFIN; ++t) for (int q = 0; q < 3; ++q) abc[3 * (k_lift + t) + q] = (double)kPolarFinish[t][q];
  return COSMO_HIP_OK;
}

namespace {

__device__ __forceinline__ long long svec_index(int i, int j) { return (long long)j * (j + 1) / 2 + i; }


// value of the (real symmetric) working matrix at (i, j) read from the cone's slice x.  Real cones: svec / square layouts.
// Complex Hermitian cone H = A + iB of side r (src/convexset.jl:444-458): the 2r x 2r embedding [[A, -B], [B, A]].
__device__ __forceinline__ real polar_read(const real* __restrict__ x, int kind, int d, int i, int j) {
  const real isq2 = 0.70710678118654752440;
  if (kind == COSMO_HIP_PSD_TRIANGLE) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    const real t = x[svec_index(a, b)];
    return (a == b) ? t : isq2 * t;
  }
  if (kind == COSMO_HIP_PSD_SQUARE) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    return (x[(long long)b * d + a] + x[(long long)a * d + b]) / R(2.0);     // symmetrize_upper! (src/algebra.jl:201-208)
  }
  const int r = d / 2;
  const int bi = i >= r, bj = j >= r, ii = i - bi * r, jj = j - bj * r;
  const int a = ii < jj ? ii : jj, b = ii < jj ? jj : ii;
  if (bi == bj) { const real t = x[svec_index(a, b)]; return (a == b) ? t : isq2 * t; }
  if (ii == jj) return 0.0;
  real im = isq2 * x[(long long)r * (r + 1) / 2 + (long long)b * (b - 1) / 2 + a];   // B[a, b] for a < b
  if (ii > jj) im = -im;                                                               // B[ii, jj]
  return (bi == 1) ? im : -im;                                                         // lower-left block B, upper-right block -B
}
// store the projected value v of the upper-triangle position (i <= j) into the cone's layout
__device__ __forceinline__ void polar_write(real* __restrict__ x, int kind, int d, int i, int j, real v) {
  const real sq2 = 1.41421356237309504880;
  if (kind == COSMO_HIP_PSD_TRIANGLE) { x[svec_index(i, j)] = (i == j) ? v : sq2 * v; return; }
  if (kind == COSMO_HIP_PSD_SQUARE) { x[(long long)j * d + i] = v; x[(long long)i * d + j] = v; return; }
  const int r = d / 2;
  if (j < r) { x[svec_index(i, j)] = (i == j) ? v : sq2 * v; return; }            // real part (extract_upper_triangle!, :474-490)
  if (i < r) { const int jj = j - r; if (i < jj) x[(long long)r * (r + 1) / 2 + (long long)jj * (jj - 1) / 2 + i] = sq2 * (-v); }   // M[i, r + jj] = -B[i, jj]
}

// X (full symmetric, zero padded to ld) from the svec / square slice of s, and the partial sums of ||X||_F^2
__global__ __launch_bounds__(COSMO_BS) void k_polar_populate(const Ctl* __restrict__ ctl, int guard, PolarCone cn, const real* __restrict__ s,
                                                             real* __restrict__ X, real* __restrict__ parts) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real* x = s + cn.off;
  const int d = cn.d, ld = cn.ld;
  real acc = 0.0;
  for (int j = blockIdx.x; j < ld; j += gridDim.x) {
    for (int i = threadIdx.x; i < ld; i += COSMO_BS) {
      real v = 0.0;
      if (i < d && j < d) { v = polar_read(x, cn.kind, d, i, j); acc += v * v; }
      X[(long long)j * ld + i] = v;
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = acc;
}

// U = 2 X / ||X||_F  (U = 0 for X = 0): spectrum in [0, 2], the domain of the lifting polynomial
__global__ __launch_bounds__(COSMO_BS) void k_polar_scale(const Ctl* __restrict__ ctl, int guard, long long n, int nparts, const real* __restrict__ parts,
                                                          const real* __restrict__ X, real* __restrict__ U, real* __restrict__ nrm_out) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real nf = sqrt(reduce_partials_sum(parts, nparts, red));
  const real inv = (nf > R(0.0)) ? R(2.0) / nf : R(0.0);
  if (blockIdx.x == 0 && threadIdx.x == 0) *nrm_out = nf;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) U[i] = X[i] * inv;
}

// C = alpha * (A B) + beta * Cin on the upper TS x TS tiles, mirrored into the lower ones.  A, B symmetric, leading dimension
// ld (multiple of TS), zero padded.  EPI 0: C = A B.  TS = 64 or 96: the host picks the tile side per cone so that the number of
// upper tiles quantises well against the 256 CUs (d = 2000: 231 tiles of 96 -> one tile per CU, instead of 528 tiles of 64
// -> three rounds on some CUs).  4 waves per workgroup, each owns a (TS/2) x (TS/2) quadrant = (TS/32)^2 MFMA accumulators.
// Pipeline: the global loads of k-panel kb+2 are issued before the MFMAs of panel kb, panel kb+1 (loaded one step earlier) is
// written to the other LDS buffer after them -- two panels of MFMA work cover the load latency per workgroup.
// Epilogue: the accumulators are transposed through LDS so that both the natural and the mirrored tile leave as full rows.
template <int TS> struct GemmCfg {
  static constexpr int NM = TS / 32;                 // MFMA tiles per wave and dimension
  static constexpr int NL = TS / 32;                 // real2 loads per thread, operand and panel
  static constexpr int PITCH = TS + 16;              // == 16 (mod 32): conflict-free ds_read_b64 of 16-lane x 4-row fragments
  static constexpr int CPITCH = TS + 1;              // odd: conflict-free column reads in the epilogue
  static constexpr int PANEL = PK * PITCH;           // doubles per operand panel
  static constexpr int SMEM = (4 * PANEL > TS * CPITCH ? 4 * PANEL : TS * CPITCH) * (int)sizeof(real);
  static constexpr int SMEM2 = (8 * PANEL > TS * CPITCH ? 8 * PANEL : TS * CPITCH) * (int)sizeof(real);   // split-k: two groups of panels
};

// SK = 2: intra-workgroup split of k.  8 waves; waves 0-3 (group 0) take the even k-panels, waves 4-7 (group 1) the odd ones, each
// group with its own double-buffered LDS panels; the two partial tiles are added through LDS before the epilogue.  Used when the
// tile count gives one tile per CU: two waves per SIMD then cover each other's LDS / barrier stalls (one wave per SIMD issues
// MFMAs only ~59 % of the time).
// Main loop of one tile: acc += A(:, i-block)' panels times B(:, j-block) panels over the k-panels [kb0, kb0 + nk) (kb0 != 0 only with
// SK = 1: the stream-K kernel).  `qact`: this wave's quadrant is computed (quadrant masking, see symm_gemm_tile).
// Fragment reads of the main loops: ONE ds_read_b64 per fragment.  Left to itself the compiler pairs the reads of two k-steps into ds_read2st64_b64, and on
// gfx950 a ds_read2_b64 is two accesses at HALF the LDS rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table: 128 against 256 bytes per clock).  A volatile
// access in the LDS address space is not merged (and stays a ds_read, not a flat load): the d = 2000 product 166.8 -> 162.1 us, the batched product of config 5
// 33.1 -> 32.3-32.9 us, same values (tools/product_lab_both.py; -DPOLAR_LAB_LDS_MERGE restores the plain loads).
#ifndef POLAR_LAB_LDS_MERGE
#define FRAG_RD(p) (*(const volatile __attribute__((address_space(3))) real*)(p))
#else
#define FRAG_RD(p) (*(p))
#endif
template <int TS, int SK>
__device__ __forceinline__ void symm_mainloop(const real* __restrict__ A, const real* __restrict__ B, int ld, int i0, int j0, bool qact, int kb0, int nk,
                                              real* smem, v4d (&acc)[TS / 32][TS / 32]) {
  using Cfg = GemmCfg<TS>;
  constexpr int NM = Cfg::NM, NL = Cfg::NL, PITCH = Cfg::PITCH, PANEL = Cfg::PANEL;
  const int grp = (SK == 2) ? (threadIdx.x >> 8) : 0;          // k-split group of this wave
  const int gtid = threadIdx.x & 255;                          // thread index within the group
  real* As = smem + grp * 4 * PANEL;   // [2][PANEL] per group
  real* Bs = As + 2 * PANEL;           // [2][PANEL]
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  const int wi = wv & 1, wj = wv >> 1;
  const long long pstep = (long long)PK * ld * SK;             // this group's next panel
  // global -> LDS mapping: real2 number q = tid + 256 u of the PK x TS panel: k = q / (TS/2), index pair = q % (TS/2)
  int goff[NL], soff[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int q = gtid + 256 * u;
    const int k = q / (TS / 2), c2 = q % (TS / 2);
    goff[u] = k * ld + 2 * c2;
    soff[u] = k * PITCH + 2 * c2;
  }
  const real* ga = A + i0 + ((long long)grp + kb0) * PK * ld;        // group 1 starts at panel 1
  const real* gb = B + j0 + ((long long)grp + kb0) * PK * ld;
  real2 r[POLAR_PREFETCH_DEPTH][2 * NL];
#ifndef POLAR_LAB_KB
#define POLAR_LAB_KB(k) (k)
#define POLAR_LAB_SYNC() __syncthreads()
#endif
#define P_LOAD(R, KB)                                                                                     \
  {                                                                                                       \
    const long long o_ = (long long)(POLAR_LAB_KB(KB)) * pstep;                                                         \
    _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                      \
      (R)[u] = *reinterpret_cast<const real2*>(ga + o_ + goff[u]);                                      \
      (R)[NL + u] = *reinterpret_cast<const real2*>(gb + o_ + goff[u]);                                 \
    }                                                                                                     \
  }
#define P_STORE(R, BUF)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                      \
      real* pa_ = As + (BUF) * PANEL + soff[u];                                                         \
      real* pb_ = Bs + (BUF) * PANEL + soff[u];                                                         \
      pa_[0] = (R)[u].x; pa_[1] = (R)[u].y;                                                               \
      pb_[0] = (R)[NL + u].x; pb_[1] = (R)[NL + u].y;                                                     \
    }                                                                                                     \
  }
  const int fa = (TS / 2) * wi + (lane & 15), fb = (TS / 2) * wj + (lane & 15), fk = lane >> 4;
#define P_COMPUTE(BUF)                                                                                    \
  if (qact) {                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < PK / 4; ++ks) {                                               \
      const real* ap = As + (BUF) * PANEL + (ks * 4 + fk) * PITCH + fa;                                 \
      const real* bp = Bs + (BUF) * PANEL + (ks * 4 + fk) * PITCH + fb;                                 \
      real av[NM], bv[NM];                                                                              \
      _Pragma("unroll") for (int a = 0; a < NM; ++a) { av[a] = FRAG_RD(ap + 16 * a); bv[a] = FRAG_RD(bp + 16 * a); }          \
      _Pragma("unroll") for (int a = 0; a < NM; ++a)                                                      \
        _Pragma("unroll") for (int b = 0; b < NM; ++b) acc[a][b] = MFMA_REAL(av[a], bv[b], acc[a][b]);     \
    }                                                                                                     \
  }
#if POLAR_PREFETCH_DEPTH == 3
  // three panels in flight: at the top of step kb, LDS[kb & 1] holds panel kb, r[(kb + 1) % 3] and r[(kb + 2) % 3] hold panels kb + 1 and
  // kb + 2 (requested two and one steps ago), panel kb + 3 is requested now -- two panels of matrix work cover a load's latency
  P_LOAD(r[0], 0)
  if (1 < nk) P_LOAD(r[1], 1)
  if (2 < nk) P_LOAD(r[2], 2)
  P_STORE(r[0], 0)
  __syncthreads();
  for (int kb0 = 0; kb0 < nk; kb0 += 6) {
#pragma unroll
    for (int st_ = 0; st_ < 6; ++st_) {                       // (not `u`: the panel macros use that name)
      const int kb = kb0 + st_;
      if (kb < nk) {
        if (kb + 3 < nk) P_LOAD(r[st_ % 3], kb + 3)            // r[kb % 3] was stored to LDS one step ago: free
        P_COMPUTE(st_ & 1)
        if (kb + 1 < nk) P_STORE(r[(st_ + 1) % 3], (st_ + 1) & 1)
        POLAR_LAB_SYNC();
      }
    }
  }
#else
  P_LOAD(r[0], 0)
  if (1 < nk) P_LOAD(r[1], 1)
  P_STORE(r[0], 0)
  __syncthreads();
  // invariant at the top of step kb: LDS[kb & 1] holds panel kb, r[(kb + 1) & 1] holds panel kb + 1 (in flight)
  for (int kb = 0; kb < nk; kb += 2) {
    if (kb + 2 < nk) P_LOAD(r[0], kb + 2)
    P_COMPUTE(0)
    if (kb + 1 < nk) P_STORE(r[1], 1)
    POLAR_LAB_SYNC();
    if (kb + 1 >= nk) break;
    if (kb + 3 < nk) P_LOAD(r[1], kb + 3)
    P_COMPUTE(1)
    if (kb + 2 < nk) P_STORE(r[0], 0)
    POLAR_LAB_SYNC();
  }
#endif
#undef P_LOAD
#undef P_STORE
#undef P_COMPUTE
}

// Epilogue of one tile, first half: the accumulators go to LDS as Cs[j][i] (pitch CPITCH), transposed so that both orientations of the
// tile can leave as full rows.  SK = 2: the two k-groups' partial tiles are added here (even panels + odd panels, fixed order).
template <int TS, int SK>
__device__ __forceinline__ void symm_acc_to_lds(v4d (&acc)[TS / 32][TS / 32], real* smem) {
  using Cfg = GemmCfg<TS>;
  constexpr int NM = Cfg::NM, CPITCH = Cfg::CPITCH;
  const int grp = (SK == 2) ? (threadIdx.x >> 8) : 0;
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  const int wi = wv & 1, wj = wv >> 1;
  real* Cs = smem;
  if (SK == 2 && grp == 1) {
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
#pragma unroll
      for (int nj = 0; nj < NM; ++nj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = (TS / 2) * wi + 16 * mi + ACC_ROW(lane, q);
          const int j = (TS / 2) * wj + 16 * nj + (lane & 15);
          Cs[j * CPITCH + i] = acc[mi][nj][q];
        }
  }
  if (SK == 2) __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
#pragma unroll
      for (int nj = 0; nj < NM; ++nj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = (TS / 2) * wi + 16 * mi + ACC_ROW(lane, q);     // D row
          const int j = (TS / 2) * wj + 16 * nj + (lane & 15);             // D col
          const real v = acc[mi][nj][q];
          Cs[j * CPITCH + i] = (SK == 2) ? (v + Cs[j * CPITCH + i]) : v;   // even-panel partial + odd-panel partial (fixed order)
        }
  }
  __syncthreads();
}
// Second half: C = alpha Cs + beta Cin on the upper tile (ti <= tj), mirrored into the lower one.  Thread t owns the elements
// e = t + 256 SK k of the tile (i = e % TS, j = e / TS) in the first loop.  The Cin values are requested a CHUNK at a time (12-18
// independent loads in flight, then their stores): written as one loop, every Cin load was a separate round trip to memory in front
// of its store (load, s_waitcnt vmcnt(0), store -- 18 to 36 serial round trips per tile).
template <int EPI, int TS, int SK>
__device__ __forceinline__ void symm_store_from_lds(const real* __restrict__ Cin, real* __restrict__ C, int ld, int ti, int tj, real alpha, real beta,
                                                    real* smem) {
  constexpr int CPITCH = GemmCfg<TS>::CPITCH;
  constexpr int NE = TS * TS / (256 * SK);             // elements per thread: 36 / 18 (TS = 96), 16 (TS = 64)
  constexpr int CH = (NE > 18) ? 12 : NE;              // chunk
  static_assert(NE % CH == 0, "chunking");
  const int i0 = ti * TS, j0 = tj * TS;
  real* Cs = smem;
  const bool diag = (ti == tj);
  // natural orientation: column j0 + j, rows i0 .. i0 + TS - 1 contiguous
#pragma unroll 1
  for (int c = 0; c < NE / CH; ++c) {
    real cin[CH];
    if (EPI == 1) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int e = threadIdx.x + 256 * SK * (c * CH + k);
        const int i = e % TS, j = e / TS;
        cin[k] = (diag && i > j) ? R(0.0) : Cin[(long long)(j0 + j) * ld + i0 + i];
      }
    }
    // the chunk's LDS values are read together and the stores are predicated (a read -> wait -> store round trip per element otherwise)
    real v[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int e = threadIdx.x + 256 * SK * (c * CH + k);
      v[k] = Cs[(e / TS) * CPITCH + (e % TS)];
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int e = threadIdx.x + 256 * SK * (c * CH + k);
      const int i = e % TS, j = e / TS;
      const bool ok = !(diag && i > j);
      if (EPI == 1) { v[k] = alpha * v[k] + beta * cin[k]; if (ok) Cs[j * CPITCH + i] = v[k]; }
      if (ok) C[(long long)(j0 + j) * ld + i0 + i] = v[k];
    }
  }
  __syncthreads();
  // mirrored orientation: column i0 + i, rows j0 .. j0 + TS - 1 contiguous
#pragma unroll 1
  for (int c = 0; c < NE / CH; ++c) {
    real v[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int e = threadIdx.x + 256 * SK * (c * CH + k);
      v[k] = Cs[(e % TS) * CPITCH + (e / TS)];
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int e = threadIdx.x + 256 * SK * (c * CH + k);
      const int j = e % TS, i = e / TS;
      if (!(diag && i >= j)) C[(long long)(i0 + i) * ld + j0 + j] = v[k];
    }
  }
}

template <int EPI, int TS, int SK>
__device__ __forceinline__ void symm_gemm_tile(const real* __restrict__ A, const real* __restrict__ B, const real* __restrict__ Cin,
                                               real* __restrict__ C, int ld, int ti, int tj, real alpha, real beta, real* smem,
                                               int kext = 0) {
  // kext: extent of the k loop (a multiple of 2 PK SK; 0 = ld).  Rows / columns beyond the cone's d are zero in every operand of
  // the iteration, so the batched path stops the inner products at d rounded up to 32 instead of the tile-rounded ld: exact.
  constexpr int NM = GemmCfg<TS>::NM;
  const int i0 = ti * TS, j0 = tj * TS;
  const int wv = (threadIdx.x >> 6) & 3;
  const int wi = wv & 1, wj = wv >> 1;
  // Quadrant masking: a wave's (TS/2) x (TS/2) quadrant is skipped (no LDS fragment reads, no MFMAs; its accumulators stay zero, which
  // IS the result there) when it lies entirely beyond the cone's extent -- the edge tiles of a cone whose side is not a multiple of
  // TS -- or when it is the strictly-lower quadrant of a diagonal tile, which the epilogue never reads (it mirrors the upper one).
  // The wave still takes part in the panel loads and barriers.  On the cone mix of BASELINE config 5 this removes 1.36x of the
  // matrix-instruction work of a product (diagonal tiles: 4 -> 3 quadrants; d = 130: 24 -> 15 quadrants) and is bit-identical -- but
  // buys only 0-12 % per product (d = 65: 31.4 -> 27.7 us, d = 129: 77.4 -> 74.2, d = 192: unchanged; cfg5 unchanged): the kernel is
  // bound by its panel loads (3.7 TB/s of HBM traffic, working set of ~300 MB per launch), not by MFMA issue
  // (profiles/r02_polar_class_time.txt).
  const int qext = kext > 0 ? kext : ld;
  const bool qact = (i0 + (TS / 2) * wi < qext) && (j0 + (TS / 2) * wj < qext) && !(ti == tj && wi > wj);
  v4d acc[NM][NM];
#pragma unroll
  for (int a = 0; a < NM; ++a)
#pragma unroll
    for (int b = 0; b < NM; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  const int nk = (kext > 0 ? kext : ld) / PK / SK;   // panels per group (a multiple of 2: both groups run the same number of steps)
#ifndef POLAR_LAB_NO_MAINLOOP      // lab: epilogue only
  symm_mainloop<TS, SK>(A, B, ld, i0, j0, qact, 0, nk, smem, acc);
#endif
  symm_acc_to_lds<TS, SK>(acc, smem);
#ifndef POLAR_LAB_NO_EPILOGUE      // lab: main loop only (one store keeps the accumulators alive)
  symm_store_from_lds<EPI, TS, SK>(Cin, C, ld, ti, tj, alpha, beta, smem);
#else
  if (threadIdx.x == 0) C[(long long)j0 * ld + i0] = smem[0];
#endif
}

// one large cone: the grid walks its upper tiles
template <int EPI, int TS, int SK>
__global__ __launch_bounds__(256 * SK) void k_symm_gemm(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate, const real* __restrict__ A,
                                                   const real* __restrict__ B, const real* __restrict__ Cin, real* __restrict__ C, int ld, int ntiles,
                                                   real alpha, real beta) {
  if (guard && ctl->halt) return;
  if (gate && !*gate) return;          // fallback round whose verification already passed
  extern __shared__ real smem[];
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own 4 MB L2.  Give every
  // XCD a CONTIGUOUS range of the column-major upper-triangle tile list (a few adjacent tile columns: one shared B panel,
  // consecutive A panels) so that the k-panels its concurrent tiles stream are fetched once per XCD, not once per tile.
  const int per_xcd = gridDim.x >> 3;
  const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  // unrank the tile: column-major over the upper triangle, (ti <= tj)
  int tj = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) / 2.0);
  while ((long long)tj * (tj + 1) / 2 > t) --tj;
  while ((long long)(tj + 1) * (tj + 2) / 2 <= t) ++tj;
  const int ti = t - tj * (tj + 1) / 2;
  symm_gemm_tile<EPI, TS, SK>(A, B, Cin, C, ld, ti, tj, alpha, beta, smem);
}

// ---- stream-K variant of the large-cone product (OPT-IN: COSMO_HIP_POLAR_STREAMK=1; a measured negative result) -----------------
// One tile per workgroup quantises badly against the 256 CUs (d = 2000: 231 tiles of 96 -> 25 CUs idle, every other CU bound to the
// barrier rhythm of its one workgroup; d = 1000: 66 tiles -> a quarter of the chip).  Here the WORK -- (tile, k-panel) units in
// tile-major order -- is cut into G equal contiguous ranges, two resident workgroups of 256 threads per CU (their barriers are
// independent, so one workgroup's LDS / barrier phases are covered by the other's matrix instructions).  A range that ends inside a tile
// leaves a PARTIAL tile in the workgroup's scratch slot and raises the slot's flag; the workgroup
// that owns the tile's LAST panels adds the partials to its own in increasing-k order and runs the epilogue.  Fixed ranges => the
// summation order is fixed: results are bit-reproducible and identical on every rank of a sharded run.
// Measured (profiles/r02_streamk_lab.txt, d = 2000, 8.58 GFLOP per product): main loops alone 156 us (the 231-workgroup kernel: 177 us
// for everything), but the exchange -- every tile is split, so every workgroup ends with a partial-tile store, a flag wait, 74 KB of
// partial loads and the Cin loads, all serial round trips to memory at the very moment its CU partner does the same -- brings the
// product to 182 us; with release / acquire fences instead of write-through stores 248 us (each fence empties the XCD's L2).  Smaller
// cones lose more (d = 500: 2.7 vs 1.8 ms per projection: 9 partials per tile).  cfg4: 120.3 vs 127.2 it/s => not the default.
//  * No deadlock by construction: a workgroup takes a TICKET when it starts and derives its range from it; partials are only ever
//    awaited from lower tickets (workgroups that have already started), whatever the residency.
//  * Each workgroup walks its range from the END: the partial (head of a later tile) is produced first, so it is ready long before
//    its consumer -- which first computes its own head partial -- asks for it.
//  * XCD locality: ticket t works in the tile range of XCD t % 8 (contiguous tile columns: shared panels in that XCD's L2); ranges never
//    straddle two XCD tile ranges, so a tile's partials are exchanged by workgroups of the same ticket class.
//  * Visibility: the partial tile and the flag are written with agent-scope stores (write-through; `s_waitcnt vmcnt(0)` + workgroup barrier
//    order them) and read with agent-scope loads -- correct wherever the workgroups were placed, and without the L2 write-back /
//    invalidate of a release / acquire fence pair (which costs every other workgroup of the XCD its cached panels).
#define SKG_TS 96

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_symm_gemm_sk(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate, const real* __restrict__ A,
                                                      const real* __restrict__ B, const real* __restrict__ Cin, real* __restrict__ C, int ld, int ntiles,
                                                      int ncls, real alpha, real beta, real* __restrict__ scratch, SkSync* __restrict__ sync, unsigned base, unsigned epoch) {
  constexpr int TS = SKG_TS, NM = TS / 32, CPITCH = GemmCfg<TS>::CPITCH;
  extern __shared__ real smem[];
  __shared__ unsigned s_ticket;
  // every workgroup of every launch takes a ticket (also when the launch is halted / gated off): the counter then advances by exactly
  // gridDim.x per launch and the host knows each launch's first ticket (`base`, modulo 2^32)
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&sync->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (guard && ctl->halt) return;
  if (gate && !*gate) return;          // fallback round whose verification already passed
  const unsigned G = gridDim.x, per = G / (unsigned)ncls;
  const unsigned t = s_ticket - base;                              // 0 .. G - 1 in arrival order
  const unsigned x = t % (unsigned)ncls, w = t / (unsigned)ncls;   // workgroup w of ticket class x (ncls = 8: class = XCD of an in-order dispatch)
  const int nk = ld / PK;
  const int T0 = (int)((long long)ntiles * x / ncls), T1 = (int)((long long)ntiles * (x + 1) / ncls);
  const long long U = (long long)(T1 - T0) * nk;               // units of this class
  const long long u0 = U * w / per, u1 = U * (w + 1) / per;
  const int wv = (threadIdx.x >> 6) & 3, wi = wv & 1, wj = wv >> 1;
  real* myslot = scratch + (size_t)(x * per + w) * (TS * TS);
  long long u = u1;
  while (u > u0) {
    const int tl = (int)((u - 1) / nk);                        // tile (relative to T0) of the last unit not yet done
    const long long tb = (long long)tl * nk;
    const int ke = (int)(u - tb);                              // segment = panels [kb, ke) of that tile
    const int kb = (int)((u0 > tb ? u0 : tb) - tb);
    const int tt = T0 + tl;
    int tj = (int)((sqrt(8.0 * (double)tt + 1.0) - 1.0) / 2.0);
    while ((long long)tj * (tj + 1) / 2 > tt) --tj;
    while ((long long)(tj + 1) * (tj + 2) / 2 <= tt) ++tj;
    const int ti = tt - tj * (tj + 1) / 2;
    const bool qact = !(ti == tj && wi > wj);
    v4d acc[NM][NM];
#pragma unroll
    for (int a = 0; a < NM; ++a)
#pragma unroll
      for (int b = 0; b < NM; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
    symm_mainloop<TS, 1>(A, B, ld, ti * TS, tj * TS, qact, kb, ke - kb, smem, acc);
    symm_acc_to_lds<TS, 1>(acc, smem);                         // Cs[j][i]; thread t owns the elements e = t + 256 k below
    real* Cs = smem;
#ifdef POLAR_LAB_NOXCHG
    if (ke < nk) {} else if (kb > 0) {} else
#endif
    if (ke < nk) {
      // producer: the partial tile leaves as TS x TS contiguous values, written THROUGH to memory (agent-scope stores), then the flag
#pragma unroll 4
      for (int k = 0; k < TS * TS / 256; ++k) { const int e = threadIdx.x + 256 * k; __hip_atomic_store(myslot + e, Cs[(e / TS) * CPITCH + e % TS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&sync->flag[x * per + w], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (kb > 0) {
        // consumer: the earlier panels of this tile were computed by the workgroups w - 1, w - 2, ... of the same class
        unsigned wf = w;
        while (wf > 0 && U * wf / per > tb) --wf;              // wf: the workgroup whose range contains the tile's first panel
        for (unsigned wp = wf; wp < w; ++wp) {
          if (U * (wp + 1) / per == U * wp / per) continue;      // empty range (more workgroups than units): nothing to await
          if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(&sync->flag[x * per + wp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > (1L << 21)) { __hip_atomic_store(&sync->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
          }
          __syncthreads();
          const real* ps = scratch + (size_t)(x * per + wp) * (TS * TS);
          // fixed order: own panels, then the partials by increasing k (agent-scope loads: served by memory, never by a stale cache line)
          // 12 loads of a thread in flight at a time (each is a round trip to memory)
#pragma unroll 1
          for (int c = 0; c < TS * TS / 256 / 12; ++c) {
            real pv[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) pv[k] = __hip_atomic_load(ps + threadIdx.x + 256 * (c * 12 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int e = threadIdx.x + 256 * (c * 12 + k); Cs[(e / TS) * CPITCH + e % TS] += pv[k]; }
          }
        }
        __syncthreads();
      }
      symm_store_from_lds<EPI, TS, 1>(Cin, C, ld, ti, tj, alpha, beta, smem);
    }
    __syncthreads();                                           // the next segment's panels reuse this LDS image
    u = tb + kb;
  }
}

// a batch of mid-size cones: one workgroup per (cone, upper tile) descriptor; buffers 0..3 of a cone are X, U/Y, Y/U, T
// OCC = waves per SIMD the register allocation targets: 3 (164 VGPRs, no spills; three workgroups per CU) or 4 (126 VGPRs, 8 spilled to
// 20 bytes of scratch; four workgroups per CU -- the 40 KB of LDS per workgroup allow exactly four).  Measured on BASELINE config 5
// (PMC: the product kernel is 31-36 % MFMA-busy and moves 3.7 TB/s, i.e. bound by neither): OCC = 4 is SLOWER, 50.9 vs 47.6 us per
// product, 147.4 vs 151.8 it/s -- the default stays 3, COSMO_HIP_POLAR_BATCH_OCC=4 selects the other instantiation.
template <int EPI, int OCC, int TS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_symm_gemm_batch(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate, const int4* __restrict__ tiles,
                                                         const BatchCone* __restrict__ cones, real* __restrict__ W, int ia, int ib, int icin, int ic,
                                                         real alpha, real beta) {
  if (guard && ctl->halt) return;
  extern __shared__ real smem[];
  const int4 td = tiles[blockIdx.x];
  if (td.x < 0) return;                      // padding of the XCD-interleaved tile list
  if (gate && !gate[td.x]) return;           // fallback round: only the cones whose verification failed (per cone, so that a cone's
                                             // arithmetic never depends on which other cones share its batch / its rank)
  const BatchCone bc = cones[td.x];
  const long long n2 = (long long)bc.ld * bc.ld;
  real* base = W + bc.woff;
#ifdef POLAR_LAB_ALIAS_READS       // lab: every cone READS the first cone's buffers (operands L2-resident), writes its own result
  const real* rbase = W;
#else
  const real* rbase = base;
#endif
  symm_gemm_tile<EPI, TS, 1>(rbase + ia * n2, rbase + ib * n2, rbase + icin * n2, base + ic * n2, bc.ld, td.y, td.z, alpha, beta, smem,
                             ((bc.d + 31) / 32) * 32);
}

// ---- ragged, block-balanced form of the batched product (default; COSMO_HIP_POLAR_BATCH_RAGGED=0 restores k_symm_gemm_batch) -------
// k_symm_gemm_batch gives every wave of a workgroup a fixed 32 x 32 quadrant of a 64 x 64 tile.  The time of a tile is then the time of
// its busiest wave: the strictly-lower quadrant of a diagonal tile and the quadrants beyond the cone's side can be skipped (quadrant
// masking), but the three / two / one remaining waves still do four 16 x 16 blocks each, wave w of every workgroup sits on SIMD w, and
// the product runs at the pace of the fullest SIMD -- which is why masking 1.36x of the matrix instructions away bought nothing, and why
// the cost of a cone was a staircase in ceil(d / 64) with 2.2x more performed than useful flops on the BASELINE config 5 mix.
// Here a cone's side is rounded to 16 (the MFMA tile), cut into ceil(d16 / 64) nearly equal parts of 1-4 blocks, and a tile is the LIST
// of its 16 x 16 blocks -- the upper blocks only on the diagonal -- dealt round-robin to the four waves (block q -> wave q % 4, slot
// q / 4): a 64 x 64 diagonal tile costs 3 slots instead of 4, a 48 x 48 off-diagonal tile 3, a 48 x 48 diagonal tile 2, and the
// k-loop stops at d16.  A wave's accumulators are its slots; the operand fragments of a slot are read from the same LDS panels at the
// block's offsets.  Same instruction, same k order per output element => bit-identical to k_symm_gemm_batch (tests/test_gpu_parity_psd.py).
// Operand panels are still loaded 64 wide from the tile's first row / column (ld stays a multiple of 64 and i0 + 64 <= ld by the way
// the parts are chosen), the epilogue is the same LDS transposition restricted to the tile's extents.
// ext = ei | ej << 8 | diag << 16 (ei, ej in blocks of 16).  The descriptor carries everything the kernel needs from the cone (ld, side, offset
// of its work matrices): one 32-byte load instead of descriptor -> cone table -> operands, i.e. one dependent round trip to memory less in a
// prologue that the in-kernel clocks (tools/ragged_timing_lab.py) put at 21 % of a tile's life.
struct alignas(32) RTile { int cone; int i0, j0; int ext; int ld, d; long long woff; };

#ifdef POLAR_LAB_TIMING       // lab instrumentation (tools/build_lab_variants.sh TIMING): shader-clock cycles of wave 0 of every workgroup, summed per phase
#define RT_MAXT 8192
__device__ unsigned long long g_rt[RT_MAXT * 5];     // per workgroup of the LAST launch: start, first panel in LDS, end of main loop, end, k-panels (plain stores: atomics
                                                      // on shared counters serialised 9 k tile ends and doubled the kernel time)
__device__ unsigned long long g_rtw[RT_MAXT];
#define RT_NOW() ((threadIdx.x == 0) ? (unsigned long long)clock64() : 0ull)
#define RT_ARG , t_first
#define RT_ADD(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < RT_MAXT) g_rt[blockIdx.x * 5 + (slot)] = (unsigned long long)(v); } while (0)
#else
#define RT_NOW() 0ull
#define RT_ADD(slot, v)
#define RT_ARG
#endif

// SC1: the operand panels are read with `sc1` buffer loads (L1 bypass, served by the XCD's L2) -- the form the persistent dependency-driven kernel needs,
// whose operands were written by OTHER CUs of the same XCD during the same launch (a plain load may hit a stale line of this CU's L1; MI355X_MICROARCH.md,
// "Inter-workgroup visibility").  16-byte sc1 loads run at the rate of plain ones.
typedef unsigned int polar_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int polar_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ real2 polar_ld_pair_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off) {      // two consecutive reals at byte_off of the buffer, sc1
#if REAL_IS_FLOAT
  const polar_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16 /* sc1 */);
  return make_real2(__uint_as_float(v[0]), __uint_as_float(v[1]));
#else
  const polar_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */);
  return make_real2(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
#endif
}
template <int NSL, int DEPTH, bool SC1, class PreLast>
__device__ __forceinline__ void symm_mainloop_r(const real* __restrict__ A, const real* __restrict__ B, int ld, int i0, int j0, int nk, real* smem,
                                                v4d (&acc)[4], const int (&oa)[4], const int (&ob)[4], PreLast pre_last, __amdgpu_buffer_rsrc_t rs, int offA, int offB
#ifdef POLAR_LAB_TIMING
                                                , unsigned long long& t_first
#endif
                                                ) {
  using Cfg = GemmCfg<64>;
  constexpr int NL = Cfg::NL, PITCH = Cfg::PITCH, PANEL = Cfg::PANEL;
  real* As = smem;
  real* Bs = As + 2 * PANEL;
  const int lane = threadIdx.x & 63;
  const long long pstep = (long long)PK * ld;
  int goff[NL], soff[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int q = threadIdx.x + 256 * u;
    const int k = q / 32, c2 = q % 32;
    goff[u] = k * ld + 2 * c2;
    soff[u] = k * PITCH + 2 * c2;
  }
  const real* ga = A + i0;
  const real* gb = B + j0;
  real2 r[DEPTH][2 * NL];          // DEPTH = 2: panels kb + 1 and kb + 2 in flight; DEPTH = 1 (the four-waves-per-SIMD build): panel kb + 1 only
#define R_LOAD(R, KB)                                                                                     \
  {                                                                                                       \
    const long long o_ = (long long)(KB) * pstep;                                                         \
    _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                      \
      if constexpr (SC1) {                                                                                \
        (R)[u] = polar_ld_pair_sc1(rs, (offA + (int)o_ + goff[u]) * (int)sizeof(real));                   \
        (R)[NL + u] = polar_ld_pair_sc1(rs, (offB + (int)o_ + goff[u]) * (int)sizeof(real));              \
      } else {                                                                                            \
        (R)[u] = *reinterpret_cast<const real2*>(ga + o_ + goff[u]);                                      \
        (R)[NL + u] = *reinterpret_cast<const real2*>(gb + o_ + goff[u]);                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
#define R_STORE(R, BUF)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                      \
      real* pa_ = As + (BUF) * PANEL + soff[u];                                                           \
      real* pb_ = Bs + (BUF) * PANEL + soff[u];                                                           \
      pa_[0] = (R)[u].x; pa_[1] = (R)[u].y;                                                               \
      pb_[0] = (R)[NL + u].x; pb_[1] = (R)[NL + u].y;                                                     \
    }                                                                                                     \
  }
  const int fl = lane & 15, fk = lane >> 4;
  const real* apb[4]; const real* bpb[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) { apb[sl] = As + fk * PITCH + fl + oa[sl]; bpb[sl] = Bs + fk * PITCH + fl + ob[sl]; }
#define R_COMPUTE(BUF)                                                                                    \
  {                                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < PK / 4; ++ks) {                                               \
      real av[NSL > 0 ? NSL : 1], bv[NSL > 0 ? NSL : 1];                                                  \
      _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                                \
        av[sl] = FRAG_RD(apb[sl] + (BUF) * PANEL + ks * 4 * PITCH);                                       \
        bv[sl] = FRAG_RD(bpb[sl] + (BUF) * PANEL + ks * 4 * PITCH);                                       \
      }                                                                                                   \
      _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) acc[sl] = MFMA_REAL(av[sl], bv[sl], acc[sl]);    \
    }                                                                                                     \
  }
  R_LOAD(r[0], 0)
  if (DEPTH == 2 && 1 < nk) R_LOAD(r[DEPTH - 1], 1)
  R_STORE(r[0], 0)
  __syncthreads();
#ifdef POLAR_LAB_TIMING
  t_first = RT_NOW();
#endif
  if (DEPTH == 1) {
    for (int kb = 0; kb + 1 < nk; kb += 2) {
      R_LOAD(r[0], kb + 1)
      R_COMPUTE(0)
      R_STORE(r[0], 1)
      __syncthreads();
      if (kb + 2 >= nk) break;
      R_LOAD(r[0], kb + 2)
      R_COMPUTE(1)
      R_STORE(r[0], 0)
      __syncthreads();
    }
  } else
  // All panels but the last run in the two-phase loop; the LAST panel is peeled so that pre_last() -- a hook for work that should overlap
  // the last panel's matrix instructions -- sits in straight-line code (inside the loop anything it loads stays live across every
  // iteration: a Cin prefetch spilled 436 bytes per lane there).
  for (int kb = 0; kb + 1 < nk; kb += 2) {
    if (kb + 2 < nk) R_LOAD(r[0], kb + 2)
    R_COMPUTE(0)
    R_STORE(r[DEPTH - 1], 1)
    __syncthreads();
    if (kb + 2 >= nk) break;
    if (kb + 3 < nk) R_LOAD(r[DEPTH - 1], kb + 3)
    R_COMPUTE(1)
    R_STORE(r[0], 0)
    __syncthreads();
  }
  pre_last();
  if ((nk - 1) & 1) R_COMPUTE(1) else R_COMPUTE(0)     // (no barrier after the last panel: the epilogue does not touch LDS)
#undef R_LOAD
#undef R_STORE
#undef R_COMPUTE
}

// one ragged tile of one product: C = alpha A B + beta Cin on the tile's blocks (the whole body of k_symm_gemm_batch_r; also called, tile after tile,
// by the persistent dependency-driven kernel k_polar_dataflow below -- the same instructions in the same order, hence the same bits)
// (Measured in round 6 and NOT kept: rotating the block lists over the waves per workgroup -- logical wave 0 carries 1.16x the mean matrix work of a BASELINE
// config 5 tile, wave 3 0.87x, and wave w of every workgroup sits on SIMD w -- 250.8 vs 250.8 it/s launch per product, 252.5 vs 255.1 in the persistent form:
// the busiest SIMD is not what bounds the product.)
// PreLast: a hook that runs before the matrix instructions of the last k-panel (the persistent kernel requests its next ticket there).
template <int EPI, int OCC, bool SC1 = false, class Hook>
__device__ __forceinline__ void ragged_tile(const RTile& td, real* __restrict__ W, int ia, int ib, int icin, int ic, real alpha, real beta, real* smem, Hook pre_last) {
  const unsigned long long t_start = RT_NOW();
  unsigned long long t_first = t_start;
  (void)t_first;
#ifdef POLAR_LAB_TIMING
  const unsigned long long w_start = (threadIdx.x == 0) ? wall_clock64() : 0ull;
#endif
  const int ld = td.ld;
  const long long n2 = (long long)ld * ld;
  real* base = W + td.woff;
  const real* A = base + ia * n2;
  const real* B = base + ib * n2;
  const real* Cin = base + icin * n2;
  real* C = base + ic * n2;
  const int ei = td.ext & 255, ej = (td.ext >> 8) & 255, diag = (td.ext >> 16) & 1;
  const int nblk = diag ? ei * (ei + 1) / 2 : ei * ej;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // this wave's blocks: q = wv, wv + 4, ...  Off-diagonal tile: q -> (q % ei, q / ei).  Diagonal tile: upper blocks column by column.
  int oa[4], ob[4];
  int nsl = 0;
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int q = wv + 4 * sl;
    int bi = 0, bj = 0;
    if (q < nblk) {
      nsl = sl + 1;
      if (diag) { bj = (q >= 6) ? 3 : (q >= 3) ? 2 : (q >= 1) ? 1 : 0; bi = q - bj * (bj + 1) / 2; }
      else { bj = q / ei; bi = q - bj * ei; }
    }
    oa[sl] = 16 * bi; ob[sl] = 16 * bj;
  }
  v4d acc[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) acc[sl] = v4d{0.0, 0.0, 0.0, 0.0};
  const int nk = ((td.d + 15) / 16);         // k-panels: rows / columns beyond d are zero in every operand of the iteration
  const int xi = 16 * ei, xj = 16 * ej;      // extents in elements
  const int i0 = td.i0, j0 = td.j0;
  // Requesting Cin before the last panel's matrix instructions (pre_last = a lambda loading it) was built and measured on BASELINE config 5: 188.1 vs
  // 188.3 it/s (three workgroups per CU, LDS epilogue), and again with the epilogue below and four workgroups per CU: 40.4 us per EPI = 1
  // product either way, 16 spilled VGPRs -- the round trip is covered by the other workgroups' main loops.
  // SC1: one buffer descriptor over the cone's four work matrices (<= 2 MB: 32-bit byte offsets), operands addressed relative to it
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, SC1 ? (int)(4 * n2 * (long long)sizeof(real)) : 0, 0x00020000);
  const int offA = (int)(ia * n2) + td.i0, offB = (int)(ib * n2) + td.j0;
#ifndef POLAR_LAB_NO_MAINLOOP               // lab builds (tools/build_lab_variants.sh): epilogue only / main loop only
  switch (nsl) {                             // wave-uniform; a wave without a block still takes part in the panel loads and barriers
    case 4: symm_mainloop_r<4, (OCC >= 4 ? 1 : 2), SC1>(A, B, ld, i0, j0, nk, smem, acc, oa, ob, pre_last, rs, offA, offB RT_ARG); break;
    case 3: symm_mainloop_r<3, (OCC >= 4 ? 1 : 2), SC1>(A, B, ld, i0, j0, nk, smem, acc, oa, ob, pre_last, rs, offA, offB RT_ARG); break;
    case 2: symm_mainloop_r<2, (OCC >= 4 ? 1 : 2), SC1>(A, B, ld, i0, j0, nk, smem, acc, oa, ob, pre_last, rs, offA, offB RT_ARG); break;
    case 1: symm_mainloop_r<1, (OCC >= 4 ? 1 : 2), SC1>(A, B, ld, i0, j0, nk, smem, acc, oa, ob, pre_last, rs, offA, offB RT_ARG); break;
    default: symm_mainloop_r<0, (OCC >= 4 ? 1 : 2), SC1>(A, B, ld, i0, j0, nk, smem, acc, oa, ob, pre_last, rs, offA, offB RT_ARG); break;
  }
#else
  (void)A; (void)B; (void)nk; (void)pre_last;
#endif
#ifdef POLAR_LAB_NO_EPILOGUE
  if (lane == 0) C[(long long)j0 * ld + i0 + wv] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];      // keeps the accumulators alive
  return;
#endif
  const unsigned long long t_main = RT_NOW();
  (void)t_main; (void)t_start;
  // Epilogue, straight from the accumulators -- no LDS, no barrier: lane l of a block holds C(i, j) for column j = l & 15 and the four rows
  // i = ACC_ROW(l, q).  Natural orientation, C(i, j) at (j0 + j) ld + i0 + i: for one q the four lanes of a column write four consecutive
  // elements (f64) -- 32-byte pieces, four instructions complete a 128-byte line in L2.  Mirrored orientation, (i0 + i) ld + j0 + j: the 16
  // lanes of a row group write 128 contiguous bytes.  On a diagonal block the elements below the diagonal come from the mirror of the ones
  // above (exactly symmetric output, as the LDS-transposed epilogue of k_symm_gemm_batch produces it).
  // Until round 3 this went through LDS (blocks -> Cs[j][i], barrier, 16 row stores, barrier, 16 mirrored stores per thread): two barriers
  // that wait for the wave with the most blocks, and an epilogue of 8.6 k (7.1 k with batched LDS reads) of a tile's 33.7 k cycles.
  (void)xi; (void)xj;
  {
    const int lj = lane & 15;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nsl) {
        const int j = ob[sl] + lj;
        const bool dblk = diag && (oa[sl] == ob[sl]);
        real v[4];
        if (EPI == 1) {
          real c[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = oa[sl] + ACC_ROW(lane, q);
            const bool ok = !(dblk && i > j);
            // Cin is exactly symmetric (every iterate is written mirrored from one value): read element (i, j) at its MIRRORED address, where the
            // 16 lanes of a row group are 128 contiguous bytes (the natural address gives 32-byte pieces)
            const real* cp = Cin + (ok ? (long long)(i0 + i) * ld + j0 + j : 0);
            c[q] = SC1 ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *cp;      // (SC1: see symm_mainloop_r)
            if (!ok) c[q] = R(0.0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = alpha * acc[sl][q] + beta * c[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = acc[sl][q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = oa[sl] + ACC_ROW(lane, q);
          if (!(dblk && i > j)) C[(long long)(j0 + j) * ld + i0 + i] = v[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = oa[sl] + ACC_ROW(lane, q);
          if (!(dblk && i >= j)) C[(long long)(i0 + i) * ld + j0 + j] = v[q];
        }
      }
    }
  }
#ifdef POLAR_LAB_TIMING
  { const unsigned long long t_end = RT_NOW();
    RT_ADD(0, t_start); RT_ADD(1, t_first); RT_ADD(2, t_main); RT_ADD(3, t_end); RT_ADD(4, nk);
    if (threadIdx.x == 0 && blockIdx.x < RT_MAXT) g_rtw[blockIdx.x] = wall_clock64() - w_start; }   // tile life on the 100 MHz constant clock: calibrates the shader clock
#endif
}

template <int EPI, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_symm_gemm_batch_r(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate,
                                                         const RTile* __restrict__ tiles, const BatchCone* __restrict__ cones, real* __restrict__ W,
                                                         int ia, int ib, int icin, int ic, real alpha, real beta) {
  extern __shared__ real smem[];
  const RTile td = tiles[blockIdx.x];        // requested TOGETHER with the halt flag (two independent scalar loads, one wait)
  const int halted = guard ? ctl->halt : 0;
  if (halted) return;
  if (td.cone < 0) return;                   // padding of the XCD-interleaved tile list
  if (gate && !gate[td.cone]) return;        // fallback round: only the cones whose verification failed
  (void)cones;
  ragged_tile<EPI, OCC>(td, W, ia, ib, icin, ic, alpha, beta, smem, []() {});
}

// ---- the MAIN SCHEDULE of the batch as ONE persistent, dependency-driven launch (round 6) ---------------------------------
// Launch-per-product form: 44 dependent launches per projection, each ~1.5 generations of tiles on 1024 workgroup slots -- the ramp and the tail of
// every launch idle most of the chip (69 % slot utilisation by the in-kernel clocks of round 3).  But product p + 1 of a cone depends only on product
// p OF THE SAME CONE.  Here every cone is pinned to an XCD (the plan does that already: all tiles of a cone share the XCD's L2), every XCD has ONE
// in-order work queue -- item q of XCD x is tile (q mod n_x) of product (q div n_x), the tile order inside a product being the cost-sorted launch
// order of the XCD's list -- and the persistent workgroups of an XCD take items by ticket.  Before a tile of product p >= 1 starts, its workgroup
// waits until the cone's completion counter has reached p x (tiles of the cone): all tiles of ALL earlier products of THAT cone are stored.  A
// finished tile is published by plain stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> counter + 1 (thread 0, at the head of its next iteration);
// the consumer reads its operands with sc1 loads, i.e. past its CU's L1 out of the XCD's L2 -- the coherence point of writers and readers, which sit
// in the same XCD by construction, whatever spills to HBM in between.  No fence, no cache invalidate.  (The first version read with plain loads
// behind `buffer_inv sc0`, what profiles/r05_xcd_resident_lab.txt had recommended: wrong -- sc0 is a workgroup-scope invalidate that leaves the L1
// alone; the streaming reads of that lab were L1-cold and could not see it.  bench/dataflow_lab.hip shows both forms.)  Deadlock-free without
// any residency assumption: an item waits only for items EARLIER in its own queue, which were taken by workgroups that are running.  A bounded
// spin turns a lost dependency into COSMO_HIP_ERR_HIP on the handle instead of a hung GPU.
struct DfProd { int ia, ib, icin, ic, epi, pad; real alpha, beta; };
#define DF_MAX_PROD 64
struct DfArgs { DfProd prod[DF_MAX_PROD]; int nprod; int xoff[9]; };
// sync layout (unsigned): [16 x] = ticket counter of XCD x (one 64-byte line each), [128] = timeout marker, [144 + c] = completed tiles of cone c
#define DF_SYNC_DONE 144
__device__ __forceinline__ unsigned df_xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7u; }
__device__ __forceinline__ unsigned df_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // sc1 load: past the L1, served by the XCD's L2
// counter read-modify-writes (agent scope; workgroup scope -- executed in the XCD's L2, where all users of a counter sit -- measured the same: 3.891 vs 3.898 ms)
#define DF_RMW(ptr) __hip_atomic_fetch_add((ptr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
template <int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_polar_dataflow(Ctl* __restrict__ ctl, int guard, const RTile* __restrict__ tiles,
                                                         const int* __restrict__ cone_nt, unsigned* __restrict__ sync, real* __restrict__ W, const DfArgs a) {
  extern __shared__ real smem[];
  __shared__ unsigned s_item;
  __shared__ int s_fail;
  if (guard && ctl->halt) return;
  const int x = (int)df_xcc_id();
  const int x0 = a.xoff[x];
  const unsigned n_x = (unsigned)(a.xoff[x + 1] - x0);
  const unsigned total = n_x * (unsigned)a.nprod;
  const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;     // SCALAR branch condition (see the note in the loop)
  if (wave0) { if (threadIdx.x == 0) { s_fail = 0; s_item = (n_x > 0) ? DF_RMW(sync + 16 * x) : 0u; } }
  int prev_cone = -1;
  unsigned nxt = 0;
  bool first = true;
  for (;;) {
    // ONE thread-0 region per iteration at the loop head, every loop exit decided on SCALAR values.  With a thread-0 region at the tail AND one at the
    // head (or a per-lane exit condition) the compiler threads the `threadIdx.x == 0` test across the back edge into an exec-masked loop NEST in which thread 0
    // leaves the inner loop alone while the other lanes of its wave run on to the barrier: the waves' barrier counts diverge and the launch hangs
    // (bench/dataflow_lab.hip reproduced it; the ISA of this form has one loop around the barriers).
    if (wave0) {
      if (threadIdx.x == 0) {
        // the tile of the previous iteration is complete (s_waitcnt vmcnt(0) + workgroup barrier at the end of the body): publish it; the ticket of this
        // iteration was requested before the last k-panel of that tile (pre_last hook) and has arrived with the s_waitcnt
        if (prev_cone >= 0) (void)DF_RMW(sync + DF_SYNC_DONE + prev_cone);
        if (!first) s_item = nxt;
      }
    }
    first = false;
    __syncthreads();
    const unsigned item = __builtin_amdgcn_readfirstlane(s_item);
    if (item >= total) break;
    const unsigned p = item / n_x, t = item - p * n_x;
    const RTile td = tiles[x0 + t];
    if (p > 0) {
      if (wave0) {
        if (threadIdx.x == 0) {
          const unsigned target = p * (unsigned)cone_nt[td.cone];
          const unsigned* cnt = sync + DF_SYNC_DONE + td.cone;
          long sp = 0;
          while (df_ld(cnt) < target) {
            __builtin_amdgcn_s_sleep(1);
            ++sp;
            if ((sp & 255) == 0 && df_ld(sync + 128)) { s_fail = 1; break; }             // somebody else gave up: leave at once
            if (sp > (1L << 20)) { s_fail = 1; __hip_atomic_store(sync + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ctl->error = COSMO_HIP_ERR_HIP; break; }   // ~1 s: a legitimate wait is a few tile lives (tens of us)
          }
        }
      }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_fail)) break;
    }
    // the operands were written by other CUs of this XCD during this launch: they are read with sc1 loads (L1 bypass), never through this CU's L1
    const DfProd pr = a.prod[p];
    auto hook = [&]() { if (wave0) { if (threadIdx.x == 0) nxt = DF_RMW(sync + 16 * x); } };      // the NEXT ticket: its round trip hides behind the last panel and the epilogue
    if (pr.epi) ragged_tile<1, OCC, true>(td, W, pr.ia, pr.ib, pr.icin, pr.ic, pr.alpha, pr.beta, smem, hook);
    else ragged_tile<0, OCC, true>(td, W, pr.ia, pr.ib, pr.icin, pr.ic, pr.alpha, pr.beta, smem, hook);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's stores have reached the L2 (write-through L1); thread 0: the ticket has arrived
    __syncthreads();                                     // ... and so have everybody's; also protects the LDS panels and s_item against the next item
    prev_cone = td.cone;
  }
}
#ifdef POLAR_LAB_TIMING
}  // namespace
extern "C" void cosmo_dbg_ragged_timing(unsigned long long* out /* RT_MAXT * 5 */) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt), sizeof(unsigned long long) * RT_MAXT * 5);
}
extern "C" void cosmo_dbg_ragged_wall(unsigned long long* out /* RT_MAXT */) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rtw), sizeof(unsigned long long) * RT_MAXT);
}
namespace {
#endif

// ---- wave-per-tile variant of the batched product (barrier-free, LDS-free main loop) ---------------------------------------
// The workgroup-per-tile kernel spends a workgroup barrier, a register -> LDS copy of two operand panels and LDS fragment reads on every
// 16-deep k-panel; with the short k-loops of the mid-size cones (9-13 panels) its main loops reach 0.67 of the sustained matrix rate and do
// not speed up when the operands are L2-resident (profiles/r02_mfma_ceiling_and_gemm_lab.txt).  Here ONE WAVE owns one 64 x 64 tile:
//   * the matrix-instruction operands are loaded straight from global memory in fragment layout -- lane l reads X[k0 + (l >> 4)][c0 + (l & 15)],
//     four full 128-byte rows per instruction -- three k-steps ahead of their use (ring of four register sets), so the loop has no barrier and no LDS;
//   * operand roles are swapped (first operand from B's panel, second from A's): a lane then holds C(j, i) with i = its low lane bits, i.e.
//     the natural-orientation stores are four full 128-byte rows per instruction without a transposition;
//   * the mirrored tile leaves through a 16 x 64 LDS strip, one 512-byte row per store;
//   * the strictly-lower 16 x 16 blocks of a diagonal tile are skipped (a second, straight-line copy of the loop).
// The k-sums run in the same order with the same instruction, so the results are bit-identical to k_symm_gemm_batch.
// Measured (profiles/r02_batch_wave_kernel.txt): NOT faster -- 50.7 vs 47.3 us per product on BASELINE config 5.  Two main loops that share
// neither barriers nor LDS staging land on the same time, and neither speeds up with L2-resident operands: the 64 x 64 tile itself (8 flop per
// operand byte moved from L2 to the CU, 9 TB/s of L2 -> L1 traffic per product) is what bounds the batched path, not its pipeline.
#define WT_DEPTH 4
#define WT_SPITCH 66
template <int EPI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_symm_gemm_batch_w(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate,
                                                         const int4* __restrict__ tiles, const BatchCone* __restrict__ cones, real* __restrict__ W,
                                                         int ia, int ib, int icin, int ic, real alpha, real beta) {
  if (guard && ctl->halt) return;
  const int4 td = tiles[blockIdx.x];
  if (td.x < 0) return;
  if (gate && !gate[td.x]) return;
  __shared__ real strip[16 * WT_SPITCH];
  const BatchCone bc = cones[td.x];
  const int ld = bc.ld;
  const long long n2 = (long long)ld * ld;
  real* base = W + bc.woff;
  const real* __restrict__ A = base + ia * n2;
  const real* __restrict__ B = base + ib * n2;
  const real* __restrict__ Cin = base + icin * n2;
  real* __restrict__ C = base + ic * n2;
  const int ti = td.y, tj = td.z, i0 = ti * 64, j0 = tj * 64;
  const int kext = ((bc.d + 31) / 32) * 32;
  const bool diag = (ti == tj);
  const int lane = threadIdx.x, l15 = lane & 15, lk = lane >> 4;
  v4d acc[4][4];            // acc[a][b][q] = C(j = 16 a + ACC_ROW(lane, q), i = 16 b + l15)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  const real* pa = A + (long long)lk * ld + i0 + l15;
  const real* pb = B + (long long)lk * ld + j0 + l15;
  const long long kstep = 4LL * ld;
  const int nsteps = kext / 4;
  real fi[WT_DEPTH][4], fj[WT_DEPTH][4];
#define WT_LOAD(SLOT, S)                                                                  \
  {                                                                                       \
    const long long o_ = (long long)(S) * kstep;                                          \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { fi[SLOT][t_] = pa[o_ + 16 * t_]; fj[SLOT][t_] = pb[o_ + 16 * t_]; } \
  }
#define WT_MMA(SLOT, DG)                                                                  \
  {                                                                                       \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                      \
      _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                    \
        if (!(DG) || b_ <= a_) acc[a_][b_] = MFMA_REAL(fj[SLOT][a_], fi[SLOT][b_], acc[a_][b_]); \
  }
  // two straight-line loop bodies (compile-time block masks): all 16 blocks, or the 10 upper blocks of a diagonal tile.  Blocks beyond the
  // cone's extent multiply zero operands (edge tiles): a per-block run-time mask costs a branch per matrix instruction and the schedule
#define WT_LOOP(DG)                                                                       \
  {                                                                                       \
    int s0 = 0;                                                                           \
    for (; s0 + 4 < nsteps; s0 += 4) {                                                    \
      WT_LOAD(3, s0 + 3) WT_MMA(0, DG)                                                    \
      WT_LOAD(0, s0 + 4) WT_MMA(1, DG)                                                    \
      WT_LOAD(1, s0 + 5) WT_MMA(2, DG)                                                    \
      WT_LOAD(2, s0 + 6) WT_MMA(3, DG)                                                    \
    }                                                                                     \
    WT_LOAD(3, s0 + 3) WT_MMA(0, DG) WT_MMA(1, DG) WT_MMA(2, DG) WT_MMA(3, DG)            \
  }
  // nsteps = kext / 4 is a multiple of 8: groups of four k-steps, three steps of loads in flight, no branch inside a group
  WT_LOAD(0, 0) WT_LOAD(1, 1) WT_LOAD(2, 2)
  if (diag) WT_LOOP(1) else WT_LOOP(0)
#undef WT_LOOP
#undef WT_LOAD
#undef WT_MMA
  // natural orientation: C(i, j) at (j0 + j) * ld + i0 + i; one instruction covers four full rows of 16 consecutive i
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    real cin[4][4];
    if (EPI == 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 16 * a + ACC_ROW(lane, q), i = 16 * b + l15;
          cin[b][q] = (diag && i > j) ? R(0.0) : Cin[(long long)(j0 + j) * ld + i0 + i];
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 16 * a + ACC_ROW(lane, q), i = 16 * b + l15;
        real v = acc[a][b][q];
        if (EPI == 1) v = alpha * v + beta * cin[b][q];
        acc[a][b][q] = v;
        if (!(diag && i > j)) C[(long long)(j0 + j) * ld + i0 + i] = v;
      }
  }
  // mirrored orientation through the LDS strip: C(j, i) at (i0 + i) * ld + j0 + j, 64 consecutive j per row
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) strip[l15 * WT_SPITCH + 16 * a + ACC_ROW(lane, q)] = acc[a][b][q];
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int i = 16 * b + rr, j = lane;
      if (!(diag && i >= j)) C[(long long)(i0 + i) * ld + j0 + j] = strip[rr * WT_SPITCH + lane];
    }
    __syncthreads();
  }
}

template <int EPI, int TS, int SK>
static void launch_symm_gemm(cosmo_hip_handle* h, int guard, const int* gate, const real* A, const real* B, const real* Cin, real* C, int ld,
                             real alpha, real beta) {
  const int nt = ld / TS, ntiles = nt * (nt + 1) / 2;
  constexpr int smem = (SK == 2) ? GemmCfg<TS>::SMEM2 : GemmCfg<TS>::SMEM;
  static std::once_flag attr_once;         // (handles of a batch group set up their plans from several threads)
  std::call_once(attr_once, [&]() { (void)hipFuncSetAttribute((const void*)k_symm_gemm<EPI, TS, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
  hipLaunchKernelGGL((k_symm_gemm<EPI, TS, SK>), dim3(((ntiles + 7) / 8) * 8), dim3(256 * SK), smem, h->stream, h->ctl, guard, gate, A, B, Cin, C, ld,
                     ntiles, alpha, beta);
  static_cast<PolarPlan*>(h->psd_polar)->launches[TS == 64 ? 0 : (SK == 2 ? 2 : 1)] += 1;
}
template <int EPI>
static void launch_symm_gemm_sk(cosmo_hip_handle* h, int guard, const int* gate, const real* A, const real* B, const real* Cin, real* C, int ld,
                                real alpha, real beta, int G, int ncls) {
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  const int nt = ld / SKG_TS, ntiles = nt * (nt + 1) / 2;
  constexpr int smem = GemmCfg<SKG_TS>::SMEM;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&]() { (void)hipFuncSetAttribute((const void*)k_symm_gemm_sk<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
  q->sk_epoch += 1u; if (q->sk_epoch == 0u) q->sk_epoch = 1u;
  hipLaunchKernelGGL((k_symm_gemm_sk<EPI>), dim3(G), dim3(256), smem, h->stream, h->ctl, guard, gate, A, B, Cin, C, ld, ntiles, ncls, alpha, beta,
                     q->sk_scratch, q->sk_sync, q->sk_base, q->sk_epoch);
  q->sk_base += (unsigned)G;
  q->launches[2] += 1;
}
// ts: tile side; sk: 1 or 2 (intra-workgroup split of k, only with ts = 96), 3: stream-K (ts = 96; skg workgroups in skc ticket classes)
static void symm_gemm(cosmo_hip_handle* h, int guard, const int* gate, int ts, int sk, int epi, const real* A, const real* B, const real* Cin, real* C,
                      int ld, real alpha, real beta, int skg = 0, int skc = 8) {
  if (sk == 3) {
    if (epi) launch_symm_gemm_sk<1>(h, guard, gate, A, B, Cin, C, ld, alpha, beta, skg, skc); else launch_symm_gemm_sk<0>(h, guard, gate, A, B, Cin, C, ld, alpha, beta, skg, skc);
    return;
  }
  if (ts == 96 && sk == 2) { if (epi) launch_symm_gemm<1, 96, 2>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); else launch_symm_gemm<0, 96, 2>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); }
  else if (ts == 96) { if (epi) launch_symm_gemm<1, 96, 1>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); else launch_symm_gemm<0, 96, 1>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); }
  else { if (epi) launch_symm_gemm<1, 64, 1>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); else launch_symm_gemm<0, 64, 1>(h, guard, gate, A, B, Cin, C, ld, alpha, beta); }
}

// ---- verification of a large cone: ||G||_F^2 partials, then the decision (one workgroup) --------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_polar_sumsq(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate, long long n,
                                                          const real* __restrict__ G, real* __restrict__ parts) {
  if (guard && ctl->halt) return;
  if (gate && !*gate) return;
  __shared__ real red[COSMO_BS / 64];
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) { const real v = G[i]; acc += v * v; }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = acc;
}
// Spectral rescaling inside the first step (large cones).  U_0 = 2 X / ||X||_F only guarantees ||U_0||_2 <= 2, and for a d x d iterate with
// a flat spectrum the largest eigenvalue sits near 2 d^(-1/2): the first lifting steps are spent bringing the WHOLE spectrum up.  The
// first product of the first step, Y = U_0^2, gives a much tighter RIGOROUS bound for free: ||U_0||_2^2 = ||Y||_2 <= ||Y||_F, so
// g = 2 / ||Y||_F^(1/2) >= 1 and U <- g U_0, Y <- g^2 Y keep the spectrum inside [0, 2] while every eigenvalue gains the factor g
// (measured on closest-correlation iterates: 3.0-3.5 at d = 300, 3.6-3.8 at d = 600, ~ 0.75 d^(1/4); a lifting step is 3.84).  One
// reduction of Y and one elementwise pass (~40 us at d = 2000) against a 169 us product per lifting step saved: BASELINE config 4
// 47 -> 44 products, 117.2 -> 124.1 it/s, largest verified error bound 6.3e-13 -> 7.9e-15 ||X||_F (profiles/r02_cfg4_spectral_rescale.json).
__global__ __launch_bounds__(COSMO_BS) void k_polar_rescale(const Ctl* __restrict__ ctl, int guard, long long n, int nparts, const real* __restrict__ parts,
                                                            real* __restrict__ U, real* __restrict__ Y) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real yf = sqrt(reduce_partials_sum(parts, nparts, red));        // ||Y||_F >= ||U||_2^2
  real g = (yf > R(0.0)) ? R(2.0) / sqrt(yf) : R(1.0);
  if (!(g >= R(1.0)) || !(g <= REAL_MAX)) g = R(1.0);                    // never scale down; NaN / inf: leave the iterate alone
  const real g2 = g * g;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) { U[i] = U[i] * g; Y[i] = Y[i] * g2; }
}
// round = 0: verification of the main schedule (always runs); round >= 1: of a fallback round (runs only if the gate is open).
// last != 0: no further round is enqueued, a failure is recorded as unverified and the gate is closed for the next projection.
__global__ __launch_bounds__(COSMO_BS) void k_polar_decide(const Ctl* __restrict__ ctl, int guard, PolarDev* __restrict__ pd, int round, int last, int nparts,
                                                           const real* __restrict__ parts, const real* __restrict__ nrm, real tol) {
  if (guard && ctl->halt) return;
  if (round > 0 && !pd->gate) return;
  __shared__ real red[COSMO_BS / 64];
  const real g2 = reduce_partials_sum(parts, nparts, red);
  if (threadIdx.x != 0) return;
  const real nf = *nrm;
  const real err = (nf > R(0.0)) ? R(0.5) * sqrt(g2) / nf : R(0.0);
  const bool ok = !(err > tol);                // NaN fails
  pd->err_last = err;
  if (err > pd->err_max || err != err) pd->err_max = err;
  if (round == 0) pd->projections += 1; else pd->rounds += 1;
  if (ok) { pd->verified += 1; pd->gate = 0; }
  else if (last) { pd->unverified += 1; pd->gate = 0; }
  else pd->gate = 1;
}

// X+ = (X + H) / 2 written in the cone's layout (svec with sqrt(2) off-diagonals / mirrored square), trace(U) partials
__global__ __launch_bounds__(COSMO_BS) void k_polar_finish(const Ctl* __restrict__ ctl, int guard, PolarCone cn, const real* __restrict__ X,
                                                           const real* __restrict__ H, const real* __restrict__ U, real* __restrict__ s,
                                                           real* __restrict__ tparts) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  real* x = s + cn.off;
  const int d = cn.d, ld = cn.ld;
  real tr = 0.0;   // trace(U) + trace(U^2) = 2 #{lambda > 0} for a converged sign matrix (zero eigenvalues count as not positive)
  for (int j = blockIdx.x; j < d; j += gridDim.x) {
    for (int i = threadIdx.x; i <= j; i += COSMO_BS) {
      const long long o = (long long)j * ld + i;
      const real v = (X[o] + H[o]) / R(2.0);
      polar_write(x, cn.kind, d, i, j, v);
      const real u = U[o];
      tr += (i == j) ? (u + u * u) : R(2.0) * u * u;
    }
  }
  tr = block_sum(tr, red);
  if (threadIdx.x == 0) tparts[blockIdx.x] = tr;
}
__global__ __launch_bounds__(COSMO_BS) void k_polar_rank(const Ctl* __restrict__ ctl, int guard, int d, int nparts, const real* __restrict__ tparts,
                                                         int* __restrict__ rank_out, int kind) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real tr = reduce_partials_sum(tparts, nparts, red);
  (void)d;
  if (threadIdx.x == 0) *rank_out = (int)llround(tr / (kind == COSMO_HIP_PSD_TRIANGLE_COMPLEX ? R(4.0) : R(2.0)));   // Hermitian embedding doubles every eigenvalue
}

// ---- batched variants of populate / scale / finish / rank: blockIdx.y = cone of the batch -------------------------------
#define BPX 16   // workgroups per cone in the elementwise kernels (= partials per cone)
__global__ __launch_bounds__(COSMO_BS) void k_bpolar_populate(const Ctl* __restrict__ ctl, int guard, const BatchCone* __restrict__ cones,
                                                              const real* __restrict__ s, real* __restrict__ W, real* __restrict__ parts) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const BatchCone cn = cones[blockIdx.y];
  const real* x = s + cn.off;
  real* X = W + cn.woff;
  const int d = cn.d, ld = cn.ld;
  real acc = 0.0;
  for (int j = blockIdx.x; j < ld; j += gridDim.x) {
    for (int i = threadIdx.x; i < ld; i += COSMO_BS) {
      real v = 0.0;
      if (i < d && j < d) { v = polar_read(x, cn.kind, d, i, j); acc += v * v; }
      X[(long long)j * ld + i] = v;
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) parts[(size_t)blockIdx.y * 2 * BPX + blockIdx.x] = acc;
}
__global__ __launch_bounds__(COSMO_BS) void k_bpolar_scale(const Ctl* __restrict__ ctl, int guard, const BatchCone* __restrict__ cones,
                                                           const real* __restrict__ parts, real* __restrict__ W, real* __restrict__ bnrm,
                                                           const int* __restrict__ ubuf) {
  if (guard && ctl->halt) return;
  const BatchCone cn = cones[blockIdx.y];
  real nf2 = 0.0;
  for (int k = 0; k < BPX; ++k) nf2 += parts[(size_t)blockIdx.y * 2 * BPX + k];     // same order in every thread: deterministic
  const real nf = sqrt(nf2);
  const real inv = (nf > R(0.0)) ? R(2.0) / nf : R(0.0);
  if (blockIdx.x == 0 && threadIdx.x == 0) bnrm[blockIdx.y] = nf;
  const long long n2 = (long long)cn.ld * cn.ld;
  const real* X = W + cn.woff;
  real* U = W + cn.woff + (ubuf ? ubuf[blockIdx.y] : 1) * n2;       // the buffer that is `iu` at the step where this cone joins the schedule
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n2; i += (long long)gridDim.x * COSMO_BS) U[i] = X[i] * inv;
}
__global__ __launch_bounds__(COSMO_BS) void k_bpolar_finish(const Ctl* __restrict__ ctl, int guard, const BatchCone* __restrict__ cones,
                                                            real* __restrict__ W, int iu, real* __restrict__ s, real* __restrict__ parts) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const BatchCone cn = cones[blockIdx.y];
  const long long n2 = (long long)cn.ld * cn.ld;
  const real* X = W + cn.woff;
  const real* U = W + cn.woff + iu * n2;
  const real* H = W + cn.woff + 3 * n2;
  real* x = s + cn.off;
  const int d = cn.d, ld = cn.ld;
  real tr = 0.0;
  for (int j = blockIdx.x; j < d; j += gridDim.x) {
    for (int i = threadIdx.x; i <= j; i += COSMO_BS) {
      const long long o = (long long)j * ld + i;
      const real v = (X[o] + H[o]) / R(2.0);
      polar_write(x, cn.kind, d, i, j, v);
      const real u = U[o];
      tr += (i == j) ? (u + u * u) : R(2.0) * u * u;
    }
  }
  tr = block_sum(tr, red);
  if (threadIdx.x == 0) parts[(size_t)blockIdx.y * 2 * BPX + BPX + blockIdx.x] = tr;
}
__global__ void k_bpolar_rank(const Ctl* __restrict__ ctl, int guard, int n, const BatchCone* __restrict__ cones, const real* __restrict__ parts,
                              int* __restrict__ rank) {
  if (guard && ctl->halt) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  real tr = 0.0;
  for (int k = 0; k < BPX; ++k) tr += parts[(size_t)c * 2 * BPX + BPX + k];
  rank[cones[c].idx] = (int)llround(tr / (cones[c].kind == COSMO_HIP_PSD_TRIANGLE_COMPLEX ? R(4.0) : R(2.0)));
}

// batched verification: per-cone ||G||_F^2 partials (G in buffer ig), then one workgroup decides for the whole batch
__global__ __launch_bounds__(COSMO_BS) void k_bpolar_sumsq(const Ctl* __restrict__ ctl, int guard, const int* __restrict__ gate, const BatchCone* __restrict__ cones,
                                                           const real* __restrict__ W, int ig, real* __restrict__ vparts) {
  if (guard && ctl->halt) return;
  if (gate && !gate[blockIdx.y]) return;
  __shared__ real red[COSMO_BS / 64];
  const BatchCone cn = cones[blockIdx.y];
  const long long n2 = (long long)cn.ld * cn.ld;
  const real* G = W + cn.woff + ig * n2;
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n2; i += (long long)gridDim.x * COSMO_BS) { const real v = G[i]; acc += v * v; }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) vparts[(size_t)blockIdx.y * BPX + blockIdx.x] = acc;
}
__global__ __launch_bounds__(COSMO_BS) void k_bpolar_decide(const Ctl* __restrict__ ctl, int guard, PolarDev* __restrict__ pd, int* __restrict__ bgate, int round,
                                                            int last, int n, const BatchCone* __restrict__ cones, const real* __restrict__ vparts,
                                                            const real* __restrict__ bnrm, real tol_factor) {
  if (guard && ctl->halt) return;
  if (round > 0 && !pd->gate) return;
  __shared__ real red[COSMO_BS / 64];
  real emax = 0.0, nfail = 0.0, nchk = 0.0;
  for (int c = threadIdx.x; c < n; c += COSMO_BS) {
    if (round > 0 && !bgate[c]) continue;      // verified in an earlier round
    real g2 = 0.0;
    for (int k = 0; k < BPX; ++k) g2 += vparts[(size_t)c * BPX + k];
    const real nf = bnrm[c];
    const real err = (nf > R(0.0)) ? R(0.5) * sqrt(g2) / nf : R(0.0);
    const bool ok = !(err > tol_factor * cones[c].d * PSD_EPS);       // NaN fails
    bgate[c] = (!ok && !last) ? 1 : 0;
    nchk += 1.0;
    if (!ok) nfail += 1.0;
    emax = (err > emax || err != err) ? err : emax;
  }
  emax = block_max(emax, red);
  nfail = block_sum(nfail, red);
  nchk = block_sum(nchk, red);
  if (threadIdx.x != 0) return;
  pd->err_last = emax;
  if (emax > pd->err_max || emax != emax) pd->err_max = emax;
  if (round == 0) pd->projections += 1; else pd->rounds += 1;
  if (nfail == R(0.0)) { pd->verified += 1; pd->gate = 0; }
  else if (last) { pd->unverified += 1; pd->gate = 0; }
  else pd->gate = 1;
  (void)nchk;
}

}  // namespace

void polar_plan_destroy(cosmo_hip_handle* h) {
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q) return;
  if (q->W) (void)hipFree(q->W);
  if (q->parts) (void)hipFree(q->parts);
  if (q->nrm) (void)hipFree(q->nrm);
  if (q->gate_host) (void)hipHostFree(q->gate_host);
  if (q->sk_scratch) (void)hipFree(q->sk_scratch);
  if (q->sk_sync) (void)hipFree(q->sk_sync);
  if (q->d_bcones) (void)hipFree(q->d_bcones);
  if (q->d_btiles) (void)hipFree(q->d_btiles);
  if (q->d_rtiles) (void)hipFree(q->d_rtiles);
  if (q->d_rtiles_rep) (void)hipFree(q->d_rtiles_rep);
  if (q->rep_host) (void)hipHostFree(q->rep_host);
  free(q->h_rtiles);
  if (q->BW) (void)hipFree(q->BW);
  if (q->bparts) (void)hipFree(q->bparts);
  if (q->bnrm) (void)hipFree(q->bnrm);
  if (q->bgate) (void)hipFree(q->bgate);
  if (q->d_lgate) (void)hipFree(q->d_lgate);
  if (q->d_ubuf) (void)hipFree(q->d_ubuf);
  if (q->bgate_host) (void)hipHostFree(q->bgate_host);
  if (q->d_df_tiles) (void)hipFree(q->d_df_tiles);
  if (q->d_df_cone_nt) (void)hipFree(q->d_df_cone_nt);
  if (q->d_df_sync) (void)hipFree(q->d_df_sync);
  if (q->df_ev[0]) { (void)hipEventDestroy(q->df_ev[0]); (void)hipEventDestroy(q->df_ev[1]); }
  if (q->dev) (void)hipFree(q->dev);
  delete q;
  h->psd_polar = nullptr;
}

bool polar_enabled(const cosmo_hip_handle* h) { return h->psd_polar != nullptr; }

// called by psd_plan_create once the Jacobi plan (which owns the cone list) exists
int32_t polar_plan_create(cosmo_hip_handle* h) {
  polar_plan_destroy(h);
  PsdPlan* p = h->psd;
  if (!p) return COSMO_HIP_OK;
  std::vector<int> large_list, batch_list = p->polar_batch;
  bool jac = h->psd_mode == 1;                                                        // cosmo_hip_set_psd_projection(EIGEN)
  if (const char* e = getenv("COSMO_HIP_PSD_LARGE")) jac = (e[0] == 'j');            // "jacobi": keep the host-paced Jacobi path for real cones
  if (!jac) large_list = p->large;
  p->large_by_polar = !jac;
  for (int idx : p->cplx) { if (p->cones[idx].d > 256) large_list.push_back(idx); else batch_list.push_back(idx); }
  const bool use_large = !large_list.empty(), use_batch = !batch_list.empty();
  if (!use_large && !use_batch) return COSMO_HIP_OK;
  PolarPlan* q = new PolarPlan();
  h->psd_polar = q;
  if (const char* e = getenv("COSMO_HIP_POLAR_KLIFT")) q->k_lift = std::min(40, std::max(0, atoi(e)));
  if (const char* e = getenv("COSMO_HIP_POLAR_ADAPT")) q->adapt = atoi(e) ? 1 : 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_COMPACT_REPAIR")) q->compact_repair = atoi(e) ? 1 : 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_RESCALE")) q->rescale = atoi(e) ? 1 : 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_STREAMK")) q->streamk = atoi(e) ? 1 : 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_TS96")) q->batch_ts96 = atoi(e) ? 1 : 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_WAVE")) q->batch_wave = atoi(e) ? 1 : 0;
  q->speculate = (large_list.size() > 2) ? 1 : 0;          // many large cones: no pipeline drain per cone (see the header)
  if (const char* e = getenv("COSMO_HIP_POLAR_SPECULATE")) q->speculate = atoi(e) ? 1 : 0;
  HIPCHK(h, hipHostMalloc((void**)&q->gate_host, 8 * sizeof(int), hipHostMallocDefault));     // {gate, rounds, verified, unverified, projections} of PolarDev
  for (int i = 0; i < 8; ++i) q->gate_host[i] = 0;
  if (const char* e = getenv("COSMO_HIP_POLAR_ROUNDS")) q->max_rounds = std::min(8, std::max(0, atoi(e)));
  HIPCHK(h, hipMalloc((void**)&q->dev, sizeof(PolarDev)));
  HIPCHK(h, hipMemset(q->dev, 0, sizeof(PolarDev)));
  memset(&q->seen, 0, sizeof(PolarDev));
  if (use_large) {
    long long woff = 0;
    for (int idx : large_list) {
      const PsdConeDev& c = p->cones[idx];
      PolarCone pc;
      pc.idx = idx; pc.off = c.off; pc.d = c.d; pc.kind = c.kind;
      // tile side: minimise (rounds of upper tiles over the 256 CUs) x (tile work)
      long long best = -1; pc.ts = 64;
      for (int ts : {64, 96}) {
        const long long nt = (c.d + ts - 1) / ts, tiles = nt * (nt + 1) / 2;
        const long long cost = ((tiles + 255) / 256) * (long long)ts * ts;
        if (best < 0 || cost < best) { best = cost; pc.ts = ts; }
      }
      if (const char* e = getenv("COSMO_HIP_POLAR_TS")) { const int v = atoi(e); if (v == 64 || v == 96) pc.ts = v; }
      pc.ld = ((c.d + pc.ts - 1) / pc.ts) * pc.ts;
      { const long long nt = pc.ld / pc.ts; pc.sk = (pc.ts == 96 && nt * (nt + 1) / 2 <= 256) ? 2 : 1; }
      if (const char* e = getenv("COSMO_HIP_POLAR_SK")) { const int v = atoi(e); if (v == 1 || (v == 2 && pc.ts == 96)) pc.sk = v; }
      pc.skg = 0; pc.skc = 8;
      if (q->streamk) {
        // stream-K: tiles of 96 whatever the count (quantisation no longer matters); 2 workgroups per CU when the cone has the work
        pc.ts = SKG_TS; pc.sk = 3;
        pc.ld = ((c.d + pc.ts - 1) / pc.ts) * pc.ts;
        const long long nt = pc.ld / pc.ts, tiles = nt * (nt + 1) / 2, nk = pc.ld / PK;
        pc.skc = tiles >= 64 ? 8 : 1;
        const long long units_min = (tiles / pc.skc) * nk;                 // units of the smallest class
        long long per = std::min<long long>(512 / pc.skc, std::max<long long>(1, units_min / 4));   // >= 4 panels per workgroup
        if (const char* e = getenv("COSMO_HIP_POLAR_SKG")) { const long long v = atoll(e) / pc.skc; if (v >= 1 && v <= 1024 / pc.skc) per = std::min(v, std::max<long long>(1, units_min)); }
        pc.skg = (int)(per * pc.skc);
        q->sk_gmax = std::max(q->sk_gmax, pc.skg);
      }
      pc.woff = woff;
      woff += 4LL * pc.ld * pc.ld;
      q->cones.push_back(pc);
    }
    HIPCHK(h, hipMalloc((void**)&q->W, sizeof(real) * (size_t)woff));
    HIPCHK(h, hipMalloc((void**)&q->parts, sizeof(real) * 2 * COSMO_MAX_PARTIALS * q->cones.size()));
    HIPCHK(h, hipMalloc((void**)&q->nrm, sizeof(real) * q->cones.size()));
    if (q->sk_gmax > 0) {
      const size_t slot = (size_t)SKG_TS * SKG_TS;
      HIPCHK(h, hipMalloc((void**)&q->sk_scratch, sizeof(real) * slot * (size_t)q->sk_gmax));
      const size_t sb = sizeof(SkSync) + sizeof(unsigned) * (size_t)q->sk_gmax;
      HIPCHK(h, hipMalloc((void**)&q->sk_sync, sb));
      HIPCHK(h, hipMemset(q->sk_sync, 0, sb));
    }
  }
  if (use_batch) {
    long long woff = 0;
    std::vector<int4> tiles;
    for (int idx : batch_list) {
      const PsdConeDev& c = p->cones[idx];
      BatchCone bc;
      bc.off = c.off; bc.d = c.d; bc.kind = c.kind; bc.idx = idx;
      // tile side (opt-in, see PolarPlan::batch_ts96): 96 x 96 tiles stream 1.5x fewer operand bytes per flop than 64 x 64 tiles (12 vs 8
      // flop per byte) and spend fewer barriers per flop; they would be used where they do not cost padding: sides in (64, 96] (one tile instead of three) and in
      // (128, 192] (three tiles instead of six).  Everything else keeps 64 x 64 tiles.
      bc.ts = (q->batch_ts96 && ((c.d > 64 && c.d <= 96) || (c.d > 128 && c.d <= 192))) ? 96 : 64;
      bc.ld = ((c.d + bc.ts - 1) / bc.ts) * bc.ts;
      bc.woff = woff;
      woff += 4LL * bc.ld * bc.ld;
      q->bcones.push_back(bc);
    }
    // longest k first: the tiles of the biggest cones start at once, the small ones fill the tail
    std::vector<int> order(q->bcones.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return q->bcones[a].ld > q->bcones[b].ld; });
    // XCD-aware order: workgroup b runs on XCD b % 8 and each XCD has its own L2, so all tiles of a cone go to ONE XCD (they
    // share the cone's operand panels: 1.5-2.5x fewer HBM reads than tiles scattered over eight L2s); the cones are dealt to the
    // XCD with the least work so far, descriptor 8 * slot + x is the slot-th tile of XCD x, short lists end with null tiles.
    // One list per tile class (64, then 96): a product is one launch per class.
    int count[2] = {0, 0};
    for (int cls = 0; cls < 2; ++cls) {
      const int ts = cls == 0 ? 64 : 96;
      std::vector<int4> tl;
      if (getenv("COSMO_HIP_POLAR_BATCH_FLAT")) {
        for (int ci : order) {
          if (q->bcones[ci].ts != ts) continue;
          const int nt = q->bcones[ci].ld / ts;
          for (int tj = 0; tj < nt; ++tj) for (int ti = 0; ti <= tj; ++ti) tl.push_back(int4{ci, ti, tj, 0});
        }
      } else {
        std::vector<std::vector<int4>> xl(8);
        long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int ci : order) {
          if (q->bcones[ci].ts != ts) continue;
          const int nt = q->bcones[ci].ld / ts;
          int x = 0;
          for (int t = 1; t < 8; ++t) if (load[t] < load[x]) x = t;
          for (int tj = 0; tj < nt; ++tj) for (int ti = 0; ti <= tj; ++ti) xl[x].push_back(int4{ci, ti, tj, 0});
          load[x] += (long long)nt * (nt + 1) / 2 * q->bcones[ci].ld;
        }
        size_t maxlen = 0;
        for (int x = 0; x < 8; ++x) maxlen = std::max(maxlen, xl[x].size());
        tl.assign(8 * maxlen, int4{-1, 0, 0, 0});
        for (int x = 0; x < 8; ++x) for (size_t sl = 0; sl < xl[x].size(); ++sl) tl[8 * sl + x] = xl[x][sl];
      }
      count[cls] = (int)tl.size();
      tiles.insert(tiles.end(), tl.begin(), tl.end());
    }
    q->nbtiles = count[0]; q->nbtiles96 = count[1];
    // ragged tile list (see k_symm_gemm_batch_r): a cone's side in blocks of 16, cut into ceil(nb16 / 4) nearly equal parts
    if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_RAGGED")) q->batch_ragged = atoi(e) ? 1 : 0;
    if (q->batch_ts96 || q->batch_wave) q->batch_ragged = 0;
    q->batch_flops_performed = q->batch_flops_useful = 0.0;
    for (const BatchCone& bc : q->bcones) q->batch_flops_useful += (double)bc.d * bc.d * (bc.d + 1.0);
    if (q->batch_ragged) {
      std::vector<std::vector<RTile>> xl(8);
      long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int ci : order) {
        const BatchCone& bc = q->bcones[ci];
        const int nb16 = (bc.d + 15) / 16, nt = (nb16 + 3) / 4;
        std::vector<int> start(nt + 1, 0);
        for (int t = 0; t < nt; ++t) start[t + 1] = start[t] + nb16 / nt + (t < nb16 % nt ? 1 : 0);   // larger parts first: i0 + 64 <= ld for every tile
        int x = 0;
        for (int t = 1; t < 8; ++t) if (load[t] < load[x]) x = t;
        for (int tj = 0; tj < nt; ++tj)
          for (int ti = 0; ti <= tj; ++ti) {
            const int ei = start[ti + 1] - start[ti], ej = start[tj + 1] - start[tj], dg = (ti == tj) ? 1 : 0;
            RTile rt; rt.cone = ci; rt.i0 = 16 * start[ti]; rt.j0 = 16 * start[tj]; rt.ext = ei | (ej << 8) | (dg << 16);
            rt.ld = bc.ld; rt.d = bc.d; rt.woff = bc.woff;
            if (rt.i0 + 64 > bc.ld || rt.j0 + 64 > bc.ld) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "ragged tile exceeds the leading dimension");
            xl[x].push_back(rt);
            const int nblk = dg ? ei * (ei + 1) / 2 : ei * ej, slots = (nblk + 3) / 4;
            load[x] += (long long)slots * nb16;
            q->batch_flops_performed += 2.0 * 256.0 * nblk * 16.0 * nb16;
          }
      }
      // Launch order inside an XCD's list = decreasing modelled cost of a tile (k-panels x slots of its fullest wave, in units of a tenth
      // of a slot, + a fixed part for prologue and epilogue): the hardware hands the next workgroup to the first free slot, i.e. it runs
      // longest-processing-time-first list scheduling if the list is sorted -- with ~1.5 tiles per slot the order of the cones (by leading
      // dimension only) left expensive tiles for the tail.  35.8 -> 32.0 us per product on BASELINE config 5 (COSMO_HIP_POLAR_BATCH_SORT=0
      // keeps the cone order; three alternating pairs of runs, profiles/r03_cfg5_ragged.txt).  A cone's tiles of one shape stay adjacent
      // (stable sort), so they still meet in the XCD's L2.
      int sort_tiles = 1;
      if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_SORT")) sort_tiles = atoi(e) ? 1 : 0;
      if (sort_tiles) {
        auto cost = [](const RTile& t) {
          const int ei = t.ext & 255, ej = (t.ext >> 8) & 255, dg = (t.ext >> 16) & 1;
          const int nblk = dg ? ei * (ei + 1) / 2 : ei * ej, slots = (nblk + 3) / 4;
          return (long long)((t.d + 15) / 16) * (10 * slots + 2) + 125;
        };
        for (int x = 0; x < 8; ++x) std::stable_sort(xl[x].begin(), xl[x].end(), [&](const RTile& a, const RTile& b) { return cost(a) > cost(b); });
      }
      size_t maxlen = 0;
      for (int x = 0; x < 8; ++x) maxlen = std::max(maxlen, xl[x].size());
      std::vector<RTile> rl(8 * std::max<size_t>(maxlen, 1), RTile{-1, 0, 0, 0, 0, 0, 0});
      for (int x = 0; x < 8; ++x) for (size_t sl = 0; sl < xl[x].size(); ++sl) rl[8 * sl + x] = xl[x][sl];
      q->nrtiles = (int)rl.size();
      { // the same per-XCD lists, contiguous, for the persistent dependency-driven launch of the main schedule
        std::vector<RTile> dl;
        std::vector<int> cnt(q->bcones.size(), 0);
        for (int x = 0; x < 8; ++x) { q->df_xoff[x] = (int)dl.size(); for (const RTile& t : xl[x]) { dl.push_back(t); cnt[(size_t)t.cone] += 1; } }
        q->df_xoff[8] = (int)dl.size();
        if (dl.empty()) dl.push_back(RTile{-1, 0, 0, 0, 0, 0, 0});
        HIPCHK(h, hipMalloc((void**)&q->d_df_tiles, sizeof(RTile) * dl.size()));
        HIPCHK(h, hipMemcpy(q->d_df_tiles, dl.data(), sizeof(RTile) * dl.size(), hipMemcpyHostToDevice));
        HIPCHK(h, hipMalloc((void**)&q->d_df_cone_nt, sizeof(int) * std::max<size_t>(cnt.size(), 1)));
        HIPCHK(h, hipMemcpy(q->d_df_cone_nt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
        HIPCHK(h, hipMalloc((void**)&q->d_df_sync, sizeof(unsigned) * (DF_SYNC_DONE + cnt.size())));
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        q->df_grid = std::max(1, prop.multiProcessorCount) * 4;
        // DEFAULT: on where every XCD's list holds at least as many tiles as the XCD has workgroup slots (BASELINE config 5: 188 tiles, 128 slots:
        // 4.035 -> 3.897 ms per ADMM iteration, same bits).  A small batch (40 cliques: ~17 tiles per XCD) has nothing to overlap -- its products are
        // already shorter than a launch ramp -- and pays a dependency wait per tile: 1.48 -> 1.79 ms, so it keeps the launch-per-product form.
        { int nmin = INT32_MAX; for (int x = 0; x < 8; ++x) nmin = std::min(nmin, q->df_xoff[x + 1] - q->df_xoff[x]); q->dataflow = (nmin >= q->df_grid / 8) ? 1 : 0; }
        if (const char* e = getenv("COSMO_HIP_POLAR_DATAFLOW")) q->dataflow = atoi(e) ? 1 : 0;
        if (const char* e = getenv("COSMO_HIP_POLAR_DATAFLOW_GRID")) { const int v = atoi(e); if (v >= 8 && v <= 8192) q->df_grid = v; }
        (void)hipFuncSetAttribute((const void*)k_polar_dataflow<4>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM); }
      HIPCHK(h, hipMalloc((void**)&q->d_rtiles, sizeof(RTile) * rl.size()));
      HIPCHK(h, hipMemcpy(q->d_rtiles, rl.data(), sizeof(RTile) * rl.size(), hipMemcpyHostToDevice));
      q->h_rtiles = malloc(sizeof(RTile) * rl.size());
      if (!q->h_rtiles) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "out of host memory");
      memcpy(q->h_rtiles, rl.data(), sizeof(RTile) * rl.size());
      HIPCHK(h, hipMalloc((void**)&q->d_rtiles_rep, sizeof(RTile) * rl.size()));
      HIPCHK(h, hipHostMalloc((void**)&q->rep_host, sizeof(RTile) * rl.size()));
      (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch_r<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
      (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch_r<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
      (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch_r<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
      (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch_r<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
    } else {
      for (const BatchCone& bc : q->bcones) { const long long nt = bc.ld / bc.ts; q->batch_flops_performed += 2.0 * (double)(nt * (nt + 1) / 2) * bc.ts * bc.ts * (((bc.d + 31) / 32) * 32); }
    }
    if (tiles.empty()) tiles.push_back(int4{-1, 0, 0, 0});
    HIPCHK(h, hipMalloc((void**)&q->BW, sizeof(real) * (size_t)woff));
    // the ragged product kernel writes a cone's d16 x d16 corner only: the padding up to ld must be zero from the start (populate / scale keep
    // it zero in X and U; Y and T are never written there), because the verification sums run over whole ld x ld buffers
    HIPCHK(h, hipMemset(q->BW, 0, sizeof(real) * (size_t)woff));
    HIPCHK(h, hipMalloc((void**)&q->d_bcones, sizeof(BatchCone) * q->bcones.size()));
    HIPCHK(h, hipMalloc((void**)&q->d_btiles, sizeof(int4) * tiles.size()));
    HIPCHK(h, hipMalloc((void**)&q->bparts, sizeof(real) * 3 * BPX * q->bcones.size()));   // norm, trace and verification partials
    HIPCHK(h, hipMalloc((void**)&q->bnrm, sizeof(real) * q->bcones.size()));
    HIPCHK(h, hipMalloc((void**)&q->bgate, sizeof(int) * q->bcones.size()));
    HIPCHK(h, hipMemset(q->bgate, 0, sizeof(int) * q->bcones.size()));
    { const size_t nb = q->bcones.size();
      { int pat = 3; if (const char* e = getenv("COSMO_HIP_POLAR_PATIENCE")) pat = std::max(1, atoi(e));      // verified projections before a cone probes one step down
        q->bk.assign(nb, q->k_lift); q->bstreak.assign(nb, 0); q->bm.assign(nb, pat); }
      HIPCHK(h, hipMalloc((void**)&q->d_lgate, sizeof(int) * nb * POLAR_LIFT_CAP));
      HIPCHK(h, hipMalloc((void**)&q->d_ubuf, sizeof(int) * nb));
      HIPCHK(h, hipHostMalloc((void**)&q->bgate_host, sizeof(int) * nb, hipHostMallocDefault));
      q->lgate_kmax = -1; }
    HIPCHK(h, hipMemcpy(q->d_bcones, q->bcones.data(), sizeof(BatchCone) * q->bcones.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(q->d_btiles, tiles.data(), sizeof(int4) * tiles.size(), hipMemcpyHostToDevice));
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<0, 3, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<1, 3, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<0, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<1, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<64>::SMEM);
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<0, 2, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<96>::SMEM);
    (void)hipFuncSetAttribute((const void*)k_symm_gemm_batch<1, 2, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<96>::SMEM);
    // Ragged kernel: FOUR workgroups per CU by default (<= 128 VGPRs: one staging set, panels requested one step ahead).  Its main loops are
    // at the pace of the matrix pipe whenever three of them share a SIMD; what the fourth wave fills is the pipe time of the others'
    // prologues and epilogues: 36.0 vs 39.0 us per product on BASELINE config 5 (four alternating pairs of runs, 166.2 vs 162.9 it/s;
    // profiles/r03_cfg5_ragged.txt).  The quadrant kernel stays at three (there four were slower, see k_symm_gemm_batch).
    q->batch_occ = q->batch_ragged ? 4 : 3;
    if (const char* e = getenv("COSMO_HIP_POLAR_BATCH_OCC")) { const int v = atoi(e); if (v == 3 || v == 4) q->batch_occ = v; }
  }
  return COSMO_HIP_OK;
}

bool polar_has_batch(const cosmo_hip_handle* h) { const PolarPlan* q = static_cast<const PolarPlan*>(h->psd_polar); return q && !q->bcones.empty(); }
bool polar_has_large(const cosmo_hip_handle* h) { const PolarPlan* q = static_cast<const PolarPlan*>(h->psd_polar); return q && !q->cones.empty(); }

// the batched product: one launch per tile class (64 x 64 tiles with the occupancy variant of the plan, then 96 x 96 tiles: their
// 74.5 KB of LDS allow two workgroups per CU)
template <int EPI>
static void launch_bgemm(PolarPlan* q, hipStream_t st, const Ctl* ctl, int guard, const int* gate, int ia, int ib, int icin, int ic, real alpha, real beta) {
  if (q->batch_ragged && q->nrep > 0) {          // compact repair launch: the failing cones' tiles only (no gate needed: the list IS the gate)
    if (q->batch_occ == 4)
      hipLaunchKernelGGL((k_symm_gemm_batch_r<EPI, 4>), dim3(q->nrep), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, (const RTile*)q->d_rtiles_rep, q->d_bcones,
                         q->BW, ia, ib, icin, ic, alpha, beta);
    else
      hipLaunchKernelGGL((k_symm_gemm_batch_r<EPI, 3>), dim3(q->nrep), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, (const RTile*)q->d_rtiles_rep, q->d_bcones,
                         q->BW, ia, ib, icin, ic, alpha, beta);
    q->repair_launch_tiles += q->nrep;
    return;
  }
  if (q->batch_ragged && q->nrtiles > 0) {
    // (the two halves of the batch on two HIP streams -- so that one half's launch tail overlaps the other's steady state -- were built and
    //  measured: 187.8-189.6 vs 187.1 it/s on BASELINE config 5, no gain, removed; profiles/r03_cfg5_ragged.txt)
    if (q->batch_occ == 4)        // default: four workgroups per CU (<= 128 VGPRs: one staging set, operand panels requested ONE step ahead)
      hipLaunchKernelGGL((k_symm_gemm_batch_r<EPI, 4>), dim3(q->nrtiles), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, (const RTile*)q->d_rtiles, q->d_bcones,
                         q->BW, ia, ib, icin, ic, alpha, beta);
    else
      hipLaunchKernelGGL((k_symm_gemm_batch_r<EPI, 3>), dim3(q->nrtiles), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, (const RTile*)q->d_rtiles, q->d_bcones,
                         q->BW, ia, ib, icin, ic, alpha, beta);
    return;
  }
  if (q->nbtiles96 > 0)
    hipLaunchKernelGGL((k_symm_gemm_batch<EPI, 2, 96>), dim3(q->nbtiles96), dim3(256), GemmCfg<96>::SMEM, st, ctl, guard, gate, q->d_btiles + q->nbtiles,
                       q->d_bcones, q->BW, ia, ib, icin, ic, alpha, beta);
  if (q->nbtiles > 0 && q->batch_wave) {
    hipLaunchKernelGGL((k_symm_gemm_batch_w<EPI>), dim3(q->nbtiles), dim3(64), 0, st, ctl, guard, gate, q->d_btiles, q->d_bcones, q->BW, ia, ib, icin, ic, alpha, beta);
  } else if (q->nbtiles > 0) {
    if (q->batch_occ == 4)
      hipLaunchKernelGGL((k_symm_gemm_batch<EPI, 4, 64>), dim3(q->nbtiles), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, q->d_btiles, q->d_bcones, q->BW,
                         ia, ib, icin, ic, alpha, beta);
    else
      hipLaunchKernelGGL((k_symm_gemm_batch<EPI, 3, 64>), dim3(q->nbtiles), dim3(256), GemmCfg<64>::SMEM, st, ctl, guard, gate, q->d_btiles, q->d_bcones, q->BW,
                         ia, ib, icin, ic, alpha, beta);
  }
}

// all mid-size cones of the batch advance together: 3 launches per step for the whole batch; with the per-cone lifting depth (PolarPlan::adapt) a
// cone joins at lifting step kmax - bk[c] and the product kernels skip its tiles before that
int32_t polar_enqueue_project_batch(cosmo_hip_handle* h, real* s, int guard) {
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  PsdPlan* p = h->psd;
  hipStream_t st = h->stream;
  const int n = (int)q->bcones.size();
  real* vparts = q->bparts + (size_t)2 * BPX * n;
  // adaptive per-cone depth needs the per-cone verification results on the host: not in speculative mode (no synchronisation inside a projection),
  // not with a schedule forced through COSMO_HIP_POLAR_KLIFT
  const bool adapt = q->adapt && !q->speculate && !getenv("COSMO_HIP_POLAR_KLIFT") && q->k_lift <= POLAR_LIFT_CAP;
  int kmax = q->k_lift;
  if (adapt) {
    kmax = 0;
    for (int c = 0; c < n; ++c) kmax = std::max(kmax, q->bk[c]);
    if (q->lgate_kmax != kmax) {                     // (a change of any bk[c] resets lgate_kmax to -1)
      std::vector<int> lg((size_t)std::max(kmax, 1) * n, 0), ub((size_t)n, 1);
      for (int c = 0; c < n; ++c) {
        const int t0 = kmax - q->bk[c];
        for (int t = t0; t < kmax; ++t) lg[(size_t)t * n + c] = 1;
        ub[(size_t)c] = (t0 & 1) ? 2 : 1;            // (iu, iy) = (1, 2) at even steps, (2, 1) at odd ones
      }
      HIPCHK(h, hipMemcpyAsync(q->d_lgate, lg.data(), sizeof(int) * lg.size(), hipMemcpyHostToDevice, st));
      HIPCHK(h, hipMemcpyAsync(q->d_ubuf, ub.data(), sizeof(int) * ub.size(), hipMemcpyHostToDevice, st));
      HIPCHK(h, hipStreamSynchronize(st));           // the host vectors go out of scope; rare (only when a depth changed)
      q->lgate_kmax = kmax;
    }
  }
  hipLaunchKernelGGL(k_bpolar_populate, dim3(BPX, n), dim3(COSMO_BS), 0, st, h->ctl, guard, q->d_bcones, s, q->BW, q->bparts);
  hipLaunchKernelGGL(k_bpolar_scale, dim3(BPX, n), dim3(COSMO_BS), 0, st, h->ctl, guard, q->d_bcones, q->bparts, q->BW, q->bnrm, adapt ? (const int*)q->d_ubuf : (const int*)nullptr);
  int iu = 1, iy = 2, products = 0;
  // main schedule as ONE persistent dependency-driven launch (k_polar_dataflow): the products are collected instead of launched; the repair rounds
  // (rare) keep the launch-per-product form
  DfArgs dfa;
  bool df = q->dataflow && q->batch_ragged && q->nrtiles > 0 && !adapt && !h->profiling && 3 * (kmax + POLAR_NFIN) + 2 <= DF_MAX_PROD;
  dfa.nprod = 0;
  auto product = [&](int epi, const int* gate, int ia, int ib, int icin, int ic, real alpha, real beta) {
    if (df) { DfProd& pr = dfa.prod[dfa.nprod++]; pr.ia = ia; pr.ib = ib; pr.icin = icin; pr.ic = ic; pr.epi = epi; pr.pad = 0; pr.alpha = alpha; pr.beta = beta; return; }
    if (epi) launch_bgemm<1>(q, st, h->ctl, guard, gate, ia, ib, icin, ic, alpha, beta);
    else launch_bgemm<0>(q, st, h->ctl, guard, gate, ia, ib, icin, ic, alpha, beta);
  };
  auto flush_dataflow = [&]() -> int32_t {
    if (!df) return COSMO_HIP_OK;
    for (int x = 0; x < 9; ++x) dfa.xoff[x] = q->df_xoff[x];
    HIPCHK(h, hipMemsetAsync(q->d_df_sync, 0, sizeof(unsigned) * (DF_SYNC_DONE + q->bcones.size()), st));
    // HIP events around the launch (two records per projection): the bench's roofline takes the launch duration from them (cosmo_hip_polar_dataflow_stats)
    if (!q->df_ev[0]) { HIPCHK(h, hipEventCreate(&q->df_ev[0])); HIPCHK(h, hipEventCreate(&q->df_ev[1])); }
    if (q->df_ev_pending && hipEventQuery(q->df_ev[1]) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, q->df_ev[0], q->df_ev[1]) == hipSuccess) { q->df_seconds += 1e-3 * (double)ms; q->df_timed += 1; }
      q->df_ev_pending = 0;
    }
    (void)hipGetLastError();
    const bool timed = !q->df_ev_pending;
    if (timed) HIPCHK(h, hipEventRecord(q->df_ev[0], st));
    hipLaunchKernelGGL((k_polar_dataflow<4>), dim3(q->df_grid), dim3(256), GemmCfg<64>::SMEM, st, h->ctl, guard, (const RTile*)q->d_df_tiles, (const int*)q->d_df_cone_nt,
                       q->d_df_sync, q->BW, dfa);
    if (timed) { HIPCHK(h, hipEventRecord(q->df_ev[1], st)); q->df_ev_pending = 1; }
    q->df_launches += 1; q->df_nprod = dfa.nprod;
    df = false;                                              // everything behind the main schedule is launched product by product
    return COSMO_HIP_OK;
  };
  auto step = [&](const real* co, const int* gate) {
    product(0, gate, iu, iu, iu, iy, 1.0, 0.0);       // Y = U^2
    product(1, gate, iy, iy, iy, 3, co[2], co[1]);    // T = c Y^2 + b Y
    product(1, gate, iu, 3, iu, iy, 1.0, co[0]);      // U' = U T + a U
    std::swap(iu, iy);
    products += 3; q->launches[3] += 3;
  };
  auto verify = [&](int round, const int* gate) -> int32_t {
    product(0, gate, iu, 0, 0, 3, 1.0, 0.0);          // H = U X = |X|
    product(1, gate, iu, 3, 0, iy, 1.0, -1.0);        // G = U H - X
    CHK(flush_dataflow());
    hipLaunchKernelGGL(k_bpolar_sumsq, dim3(BPX, n), dim3(COSMO_BS), 0, st, h->ctl, guard, gate, q->d_bcones, q->BW, iy, vparts);
    hipLaunchKernelGGL(k_bpolar_decide, dim3(1), dim3(COSMO_BS), 0, st, h->ctl, guard, q->dev, q->bgate, round, round == q->max_rounds ? 1 : 0, n, q->d_bcones,
                       vparts, q->bnrm, q->tol_factor);
    products += 2; q->launches[3] += 2;
    return COSMO_HIP_OK;
  };
  for (int t = 0; t < kmax; ++t) step(kPolarLift, adapt ? (const int*)(q->d_lgate + (size_t)t * n) : (const int*)nullptr);
  for (int t = 0; t < POLAR_NFIN; ++t) step(kPolarFinish[t], nullptr);
  CHK(verify(0, nullptr));
  q->products_last_batch = products;
  { double wsum = 0.0, w3 = 0.0;                             // what the launches cost: products of a cone weighted by its d^3
    for (int c = 0; c < n; ++c) { const double d3 = (double)q->bcones[c].d * q->bcones[c].d * q->bcones[c].d; w3 += d3; wsum += d3 * (3.0 * ((adapt ? q->bk[c] : kmax) + POLAR_NFIN) + 2.0); }
    q->wprod_last = w3 > 0.0 ? wsum / w3 : 0.0; }
  for (int r = 1; r <= q->max_rounds; ++r) {                 // guarded fallback rounds (even number of steps: iu is preserved)
    if (!q->speculate) {                                     // ask the device whether the last verification failed (and, round 1 of an adaptive run, which cones)
      HIPCHK(h, hipMemcpyAsync(q->gate_host, &q->dev->gate, (adapt && r == 1 ? 5 : 1) * sizeof(int), hipMemcpyDeviceToHost, st));
      if (r == 1 && q->bgate_host) HIPCHK(h, hipMemcpyAsync(q->bgate_host, q->bgate, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));   // per-cone results: depth control + compact repair list
      HIPCHK(h, hipStreamSynchronize(st));
      if (adapt && r == 1 && q->gate_host[4] != q->seen_proj_batch) {
        // depth control from this projection's per-cone results.  PolarDev::projections advances only when the verification really ran: in a
        // halted loop (status decided, or a Krylov stall waiting for the host) every kernel above was a no-op and the flags are stale
        q->seen_proj_batch = q->gate_host[4];
        q->depth_proj += 1;
        bool changed = false;
        for (int c = 0; c < n; ++c) {
          if (*q->gate_host && q->bgate_host[c]) {           // failed at depth bk[c]: one step deeper, probe downwards half as often
            q->depth_fail += 1;
            if (q->bk[c] < POLAR_LIFT_CAP) { q->bk[c] += 1; changed = true; }
            q->bstreak[c] = 0;
            q->bm[c] = std::min(96, 2 * q->bm[c]);
          } else if (++q->bstreak[c] >= q->bm[c] && q->bk[c] > 0) {
            q->bk[c] -= 1; q->bstreak[c] = 0; q->depth_down += 1; changed = true;
          }
        }
        if (changed) q->lgate_kmax = -1;
      }
      if (!*q->gate_host) break;
    }
    int rlift = POLAR_RLIFT;
    if (q->compact_repair && q->batch_ragged && q->h_rtiles && q->bgate_host && !q->speculate) {
      if (r == 1) {                                          // tile list of the failing cones (bgate_host was read with the gate above)
        const RTile* all = (const RTile*)q->h_rtiles; RTile* dst = (RTile*)q->rep_host;
        int cnt = 0;
        for (int t = 0; t < q->nrtiles; ++t) if (all[t].cone >= 0 && q->bgate_host[all[t].cone]) dst[cnt++] = all[t];
        if (cnt > 0) {
          HIPCHK(h, hipMemcpyAsync(q->d_rtiles_rep, dst, sizeof(RTile) * (size_t)cnt, hipMemcpyHostToDevice, st));
          q->nrep = cnt; q->repair_trains += 1;
        }
        if (adapt) rlift = 1;                                // 1 + POLAR_NFIN steps: even, the buffer parity is preserved
      }
      // round 2 (r == 2): the same list behind the per-cone gate -- the cones that failed again are a subset of it
    }
    for (int t = 0; t < rlift; ++t) step(kPolarLift, q->bgate);
    for (int t = 0; t < POLAR_NFIN; ++t) step(kPolarFinish[t], q->bgate);
    CHK(verify(r, q->bgate));
  }
  q->nrep = 0;
  hipLaunchKernelGGL(k_bpolar_finish, dim3(BPX, n), dim3(COSMO_BS), 0, st, h->ctl, guard, q->d_bcones, q->BW, iu, s, q->bparts);
  hipLaunchKernelGGL(k_bpolar_rank, dim3((n + 63) / 64), dim3(64), 0, st, h->ctl, guard, n, q->d_bcones, q->bparts, p->rank);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t polar_enqueue_project(cosmo_hip_handle* h, real* s, int guard) {
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  PsdPlan* p = h->psd;
  hipStream_t st = h->stream;
  for (size_t ci = 0; ci < q->cones.size(); ++ci) {
    const PolarCone& cn = q->cones[ci];
    const long long n2 = (long long)cn.ld * cn.ld;
    real* X = q->W + cn.woff;
    real* U = X + n2;
    real* Y = U + n2;
    real* T = Y + n2;
    real* nparts = q->parts + 2 * COSMO_MAX_PARTIALS * ci;
    real* tparts = nparts + COSMO_MAX_PARTIALS;
    const int gpop = std::min(cn.ld, 1024);
    int products = 0;
    hipLaunchKernelGGL(k_polar_populate, dim3(gpop), dim3(COSMO_BS), 0, st, h->ctl, guard, cn, s, X, nparts);
    hipLaunchKernelGGL(k_polar_scale, dim3(1024), dim3(COSMO_BS), 0, st, h->ctl, guard, n2, gpop, nparts, X, U, q->nrm + ci);
    auto step = [&](const real* co, const int* gate) {
      symm_gemm(h, guard, gate, cn.ts, cn.sk, 0, U, U, nullptr, Y, cn.ld, 1.0, 0.0, cn.skg, cn.skc);         // Y = U^2
      symm_gemm(h, guard, gate, cn.ts, cn.sk, 1, Y, Y, Y, T, cn.ld, co[2], co[1], cn.skg, cn.skc);           // T = c Y^2 + b Y
      symm_gemm(h, guard, gate, cn.ts, cn.sk, 1, U, T, U, Y, cn.ld, 1.0, co[0], cn.skg, cn.skc);             // U' = U T + a U
      std::swap(U, Y);
      products += 3;
    };
    auto verify = [&](int round, const int* gate) {
      symm_gemm(h, guard, gate, cn.ts, cn.sk, 0, U, X, nullptr, T, cn.ld, 1.0, 0.0, cn.skg, cn.skc);         // H = U X = |X|
      symm_gemm(h, guard, gate, cn.ts, cn.sk, 1, U, T, X, Y, cn.ld, 1.0, -1.0, cn.skg, cn.skc);              // G = U H - X = (U^2 - I) X
      hipLaunchKernelGGL(k_polar_sumsq, dim3(1024), dim3(COSMO_BS), 0, st, h->ctl, guard, gate, n2, Y, nparts);
      hipLaunchKernelGGL(k_polar_decide, dim3(1), dim3(COSMO_BS), 0, st, h->ctl, guard, q->dev, round, round == q->max_rounds ? 1 : 0, 1024, nparts,
                         q->nrm + ci, q->tol_factor * cn.d * PSD_EPS);
      products += 2;
    };
    const bool adapt = q->adapt && !q->speculate && !getenv("COSMO_HIP_POLAR_KLIFT") && q->k_lift <= POLAR_LIFT_CAP;
    if (adapt && q->lk.size() != q->cones.size()) { q->lk.assign(q->cones.size(), q->k_lift); q->lstreak.assign(q->cones.size(), 0); q->lm.assign(q->cones.size(), 3); }
    int k_main = adapt ? q->lk[ci] : q->k_lift;
    if (q->rescale && k_main > 0) {
      // first lifting step with the spectral rescaling between its first and second product; from d = 1024 on the gain (>= 4.2)
      // exceeds the slope of a lifting step, so the main schedule is one step shorter
      symm_gemm(h, guard, nullptr, cn.ts, cn.sk, 0, U, U, nullptr, Y, cn.ld, 1.0, 0.0, cn.skg, cn.skc);
      hipLaunchKernelGGL(k_polar_sumsq, dim3(1024), dim3(COSMO_BS), 0, st, h->ctl, guard, (const int*)nullptr, n2, Y, nparts);
      hipLaunchKernelGGL(k_polar_rescale, dim3(1024), dim3(COSMO_BS), 0, st, h->ctl, guard, n2, 1024, nparts, U, Y);
      symm_gemm(h, guard, nullptr, cn.ts, cn.sk, 1, Y, Y, Y, T, cn.ld, kPolarLift[2], kPolarLift[1], cn.skg, cn.skc);
      symm_gemm(h, guard, nullptr, cn.ts, cn.sk, 1, U, T, U, Y, cn.ld, 1.0, kPolarLift[0], cn.skg, cn.skc);
      std::swap(U, Y);
      products += 3;
      k_main -= 1;
      if (cn.d >= 1024 && k_main > 1) k_main -= 1;
    }
    for (int t = 0; t < k_main; ++t) step(kPolarLift, nullptr);
    for (int t = 0; t < POLAR_NFIN_LARGE; ++t) step(kPolarFinish[t], nullptr);
    verify(0, nullptr);
    q->products_last_large = products;
    for (int r = 1; r <= q->max_rounds; ++r) {
      if (!q->speculate) {
        HIPCHK(h, hipMemcpyAsync(q->gate_host, &q->dev->gate, (adapt && r == 1 ? 5 : 1) * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (adapt && r == 1 && q->gate_host[4] != q->seen_proj_large) {        // depth control of this cone (see PolarPlan::adapt)
          q->seen_proj_large = q->gate_host[4];
          q->depth_proj += 1;
          if (*q->gate_host) { q->depth_fail += 1; q->lk[ci] = std::min(POLAR_LIFT_CAP, q->lk[ci] + 1); q->lstreak[ci] = 0; q->lm[ci] = std::min(96, 2 * q->lm[ci]); }
          else if (++q->lstreak[ci] >= q->lm[ci] && q->lk[ci] > 2) { q->lk[ci] -= 1; q->lstreak[ci] = 0; q->depth_down += 1; }
        }
        if (!*q->gate_host) break;
      }
      for (int t = 0; t < POLAR_RLIFT_LARGE; ++t) step(kPolarLift, &q->dev->gate);
      for (int t = 0; t < POLAR_NFIN_LARGE; ++t) step(kPolarFinish[t], &q->dev->gate);
      verify(r, &q->dev->gate);
    }
    const int gfin = std::min(cn.d, 1024);
    hipLaunchKernelGGL(k_polar_finish, dim3(gfin), dim3(COSMO_BS), 0, st, h->ctl, guard, cn, X, T, U, s, tparts);
    hipLaunchKernelGGL(k_polar_rank, dim3(1), dim3(COSMO_BS), 0, st, h->ctl, guard, cn.d, gfin, tparts, p->rank + cn.idx, cn.kind);
  }
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// Host-side adaptation at a synchronisation point of the loop: when fallback rounds ran on a third or more of the projections
// since the last look, the main schedule gets three more lifting steps (sticky).  Probing DOWNWARDS was built and measured (r02): on
// BASELINE configs 4 and 5 the iterates need the default ten lifting steps within a few dozen iterations, every failed probe costs a
// fallback round (26 products), and the runs ended slower (cfg4 116 -> 97 it/s) -- so the schedule only ever grows.  Not in sharded
// runs (every rank must keep the schedule of the single-rank run so that the exchanged slices stay bit-identical to it).
int32_t polar_adapt(cosmo_hip_handle* h) {
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q || !q->dev || h->comm || getenv("COSMO_HIP_POLAR_KLIFT")) return COSMO_HIP_OK;
  if (q->adapt && !q->speculate) return COSMO_HIP_OK;          // the per-cone depth control (PolarPlan::adapt) has taken over
  PolarDev now;
  HIPCHK(h, hipMemcpyAsync(&now, q->dev, sizeof now, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int dr = now.rounds - q->seen.rounds, dp = now.projections - q->seen.projections;
  q->seen = now;
  if (dp > 0 && 3 * dr >= dp && q->k_lift < 18) q->k_lift = std::min(18, q->k_lift + 3);
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_polar_stats(cosmo_hip_handle* h, int64_t out[16]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q) return COSMO_HIP_OK;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  PolarDev now;
  HIPCHK(h, hipMemcpyAsync(&now, q->dev, sizeof now, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out[0] = (int64_t)q->cones.size(); out[1] = (int64_t)q->bcones.size();
  if (!q->cones.empty()) { out[2] = q->cones[0].ts; out[3] = q->cones[0].sk; }
  out[4] = q->launches[0]; out[5] = q->launches[1]; out[6] = q->launches[2]; out[7] = q->launches[3];
  out[8] = q->products_last_large; out[9] = now.rounds; out[10] = now.verified; out[11] = q->products_last_batch;
  out[12] = q->k_lift + POLAR_NFIN; out[13] = now.unverified; out[14] = now.projections;
  out[15] = (int64_t)llround((double)now.err_max * 1e18);       // max verified error bound relative to ||X||_F, in units of 1e-18
  return COSMO_HIP_OK;
}

// persistent dependency-driven main schedule of the batch (k_polar_dataflow): out = {enabled, launches, products per launch, event-timed launches,
// average seconds per timed launch, matrix flops performed per launch, workgroups of the launch, tiles per product}
extern "C" int32_t cosmo_hip_polar_dataflow_stats(cosmo_hip_handle* h, double out[8]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q) return COSMO_HIP_OK;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (q->df_ev_pending) {
    HIPCHK(h, hipEventSynchronize(q->df_ev[1]));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, q->df_ev[0], q->df_ev[1]) == hipSuccess) { q->df_seconds += 1e-3 * (double)ms; q->df_timed += 1; }
    q->df_ev_pending = 0;
  }
  out[0] = q->dataflow; out[1] = (double)q->df_launches; out[2] = q->df_nprod; out[3] = (double)q->df_timed;
  out[4] = q->df_timed ? q->df_seconds / (double)q->df_timed : 0.0;
  out[5] = q->batch_flops_performed * q->df_nprod; out[6] = q->df_grid; out[7] = q->df_xoff[8];
  return COSMO_HIP_OK;
}
// reset of the timing part (bench: time the launches of the timed window only)
extern "C" int32_t cosmo_hip_polar_dataflow_reset_timing(cosmo_hip_handle* h) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (q) { q->df_seconds = 0.0; q->df_timed = 0; }
  return COSMO_HIP_OK;
}

// out = {adaptive depth on, min / max / mean x 1000 of the per-cone lifting depths (batch cones, else large cones), d^3-weighted products per
//        projection x 1000 of the batch's last main schedule, failed verifications, downward probes, projections seen by the depth control}
extern "C" int32_t cosmo_hip_polar_depth_stats(cosmo_hip_handle* h, int64_t out[8]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q) return COSMO_HIP_OK;
  const bool adapt = q->adapt && !q->speculate && !getenv("COSMO_HIP_POLAR_KLIFT") && q->k_lift <= POLAR_LIFT_CAP;
  out[0] = adapt ? 1 : 0;
  const std::vector<int>& ks = !q->bk.empty() ? q->bk : q->lk;
  if (adapt && !ks.empty()) {
    long long sum = 0; int mn = ks[0], mx = ks[0];
    for (int v : ks) { sum += v; mn = std::min(mn, v); mx = std::max(mx, v); }
    out[1] = mn; out[2] = mx; out[3] = (1000 * sum) / (long long)ks.size();
  } else { out[1] = out[2] = q->k_lift; out[3] = 1000LL * q->k_lift; }
  out[4] = (int64_t)llround(1000.0 * q->wprod_last);
  out[5] = q->depth_fail; out[6] = q->depth_down; out[7] = q->depth_proj;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_polar_streamk_stats(cosmo_hip_handle* h, int64_t out[4]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = 0;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q || q->cones.empty() || !q->sk_sync) return COSMO_HIP_OK;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  unsigned hdr[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(hdr, q->sk_sync, sizeof hdr, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out[0] = 1; out[1] = q->cones[0].skg; out[2] = q->cones[0].skc; out[3] = hdr[1];
  return COSMO_HIP_OK;
}

// measurement hook (bench.py roofline): the product kernel exactly as the projection launches it
extern "C" int32_t cosmo_hip_time_psd_product(cosmo_hip_handle* h, int32_t which, int32_t reps, double* avg_seconds, double* flops) {
  if (!h || reps <= 0 || !avg_seconds) return COSMO_HIP_ERR_INVALID;
  PolarPlan* q = static_cast<PolarPlan*>(h->psd_polar);
  if (!q || (which == 0 && q->cones.empty()) || (which != 0 && q->bcones.empty())) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "time_psd_product: no such cones");
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  double fl = 0.0;
  const long long keep[4] = {q->launches[0], q->launches[1], q->launches[2], q->launches[3]};
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) HIPCHK(h, hipEventRecord(e0, h->stream));
    const int R = pass == 0 ? 2 : reps;
    for (int i = 0; i < R; ++i) {
      if (which == 0) {
        const PolarCone& cn = q->cones[0];
        const long long n2 = (long long)cn.ld * cn.ld;
        real* X = q->W + cn.woff;
        symm_gemm(h, 0, nullptr, cn.ts, cn.sk, 0, X + n2, X + n2, nullptr, X + 2 * n2, cn.ld, 1.0, 0.0, cn.skg, cn.skc);
        const long long nt = cn.ld / cn.ts;
        fl = 2.0 * (double)(nt * (nt + 1) / 2) * cn.ts * cn.ts * cn.ld;
      } else if (which == 1) {
        launch_bgemm<0>(q, h->stream, h->ctl, 0, (const int*)nullptr, 1, 1, 1, 2, 1.0, 0.0);
        fl = q->batch_flops_performed;
      } else {
        // which == 2: the IN-LOOP MIX of one step of the sign iteration (polar_enqueue_project_batch): Y = U^2 (EPI 0, one operand), T = c Y^2 + b Y
        // (EPI 1, Cin = the operand) and U' = U T + a U (EPI 1, three distinct matrices) -- two thirds of a projection's products carry the
        // alpha A B + beta Cin epilogue and read up to three matrices.  Coefficients (a, b, c) = (1, 0, 0) keep the work matrices bounded over
        // any number of repetitions (T = 0, U' = U); the kernels do the same work for any coefficients.
        launch_bgemm<0>(q, h->stream, h->ctl, 0, (const int*)nullptr, 1, 1, 1, 2, 1.0, 0.0);
        launch_bgemm<1>(q, h->stream, h->ctl, 0, (const int*)nullptr, 2, 2, 2, 3, 0.0, 0.0);
        launch_bgemm<1>(q, h->stream, h->ctl, 0, (const int*)nullptr, 1, 3, 1, 2, 1.0, 1.0);
        fl = q->batch_flops_performed;
      }
    }
    if (pass == 1) HIPCHK(h, hipEventRecord(e1, h->stream));
  }
  HIPCHK(h, hipEventSynchronize(e1));
  for (int i = 0; i < 4; ++i) q->launches[i] = keep[i];
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  *avg_seconds = (double)ms * 1e-3 / ((double)reps * (which == 2 ? 3.0 : 1.0));        // per PRODUCT
  if (flops) *flops = fl;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return COSMO_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Definiteness test of the Hermitian cones for the infeasibility certificates: is_pos_def!(X, tol) of the reference adds tol to
// the diagonal and asks whether cholesky!(Hermitian(X)) succeeds (src/algebra.jl:226-233; in_dual! / in_pol_recc!,
// src/convexset.jl:415-424).  Here the same question is put to the real symmetric embedding M = [[A, -B], [B, A]] of
// sign * H (+ tol I), which is positive definite exactly when H is: one workgroup factorises M column by column (left-looking
// Cholesky in a global scratch matrix, the current row of L cached in LDS) and reports whether every pivot was > 0.
// The certificates are evaluated every check_infeasibility iterations and only after the cheap norm conditions passed, so the
// O(d^3 / 3) of one workgroup is off the hot path.  Side 2r <= COSMO_CPLX_CHOL_MAX; larger Hermitian cones never certify.
// ---------------------------------------------------------------------------------------------------------------------
#define COSMO_CPLX_CHOL_MAX 1024
__global__ __launch_bounds__(COSMO_BS) void k_cplx_chol_pd(int off, int d, const real* __restrict__ vec, real sign, real tol,
                                                           real* __restrict__ G, int* __restrict__ ok_out) {
  __shared__ real rowj[COSMO_CPLX_CHOL_MAX];
  __shared__ real piv;
  __shared__ int fail;
  const real* x = vec + off;
  for (long long e = threadIdx.x; e < (long long)d * d; e += COSMO_BS) {
    const int i = (int)(e % d), j = (int)(e / d);
    real v = sign * polar_read(x, COSMO_HIP_PSD_TRIANGLE_COMPLEX, d, i, j);
    if (i == j) v += tol;
    G[e] = v;                                                     // column major: G[j * d + i]
  }
  if (threadIdx.x == 0) fail = 0;
  __syncthreads();
  for (int j = 0; j < d; ++j) {
    for (int k = threadIdx.x; k < j; k += COSMO_BS) rowj[k] = G[(long long)k * d + j];          // L[j, 0..j-1]
    __syncthreads();
    for (int i = j + threadIdx.x; i < d; i += COSMO_BS) {
      real v = G[(long long)j * d + i];
      for (int k = 0; k < j; ++k) v -= G[(long long)k * d + i] * rowj[k];
      G[(long long)j * d + i] = v;
      if (i == j) { piv = v; if (!(v > R(0.0))) fail = 1; }
    }
    __syncthreads();
    if (fail) break;
    const real ljj = sqrt(piv);
    for (int i = j + threadIdx.x; i < d; i += COSMO_BS) G[(long long)j * d + i] = (i == j) ? ljj : G[(long long)j * d + i] / ljj;
    __syncthreads();
  }
  if (threadIdx.x == 0) *ok_out = fail ? 0 : 1;
}

// ok[q] for the q-th entry of p->cplx: 1 = sign * H + tol I is positive definite, 0 = not (or too large to test)
int32_t polar_complex_is_pd(cosmo_hip_handle* h, const real* vec, real sign, real tol, std::vector<int>& ok) {
  PsdPlan* p = h->psd;
  ok.clear();
  if (!p || p->cplx.empty()) return COSMO_HIP_OK;
  ok.assign(p->cplx.size(), 0);
  int dmax = 0;
  for (int idx : p->cplx) if (p->cones[idx].d <= COSMO_CPLX_CHOL_MAX) dmax = std::max(dmax, p->cones[idx].d);
  if (dmax == 0) return COSMO_HIP_OK;
  real* G = nullptr; int* d_ok = nullptr;
  HIPCHK(h, hipMalloc((void**)&G, sizeof(real) * (size_t)dmax * dmax));
  if (hipMalloc((void**)&d_ok, sizeof(int)) != hipSuccess) { (void)hipFree(G); return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipMalloc failed"); }
  int32_t rc = COSMO_HIP_OK;
  for (size_t q = 0; q < p->cplx.size() && rc == COSMO_HIP_OK; ++q) {
    const PsdConeDev& cn = p->cones[p->cplx[q]];
    if (cn.d > COSMO_CPLX_CHOL_MAX) continue;
    hipLaunchKernelGGL(k_cplx_chol_pd, dim3(1), dim3(COSMO_BS), 0, h->stream, cn.off, cn.d, vec, sign, tol, G, d_ok);
    int v = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&v, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "complex definiteness test failed");
    ok[q] = v;
  }
  (void)hipFree(G); (void)hipFree(d_ok);
  return rc;
}
