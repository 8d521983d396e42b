// kernels.hip -- hand-written gfx950 kernels of the ADMM iteration: CSR-stream SpMV family, the fused vector
// phases of admm_z!/admm_x!/admm_w!, simple-cone projections, the device-resident CG, residual checks and the
// adaptive-rho rule.  Compiled with -ffp-contract=off: the elementwise phases must round exactly like Julia's
// broadcasts (no FMA), see SURVEY.md Appendix A.  Reference citations are to /root/reference/src.
#include "device_utils.h"

// ---------------------------------------------------------------------------------------------------------------------
// plain SpMV  y = M x                      (mul!(y, A, x) / mul!(y, A', x) / mul!(y, P, x), residuals.jl:4,12,15)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_spmv_plain(CsrView M, const real* __restrict__ x, real* __restrict__ y) {
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  for (int b = blockIdx.x; b < M.nb; b += gridDim.x) {
    csr_stream_tile(M, x, x, tile_of_block(b, M.nb, M.xcd_affine), lds, red, [&](int r, real s1, real s2) { y[r] = s1 + s2; });
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// admm_z! simple part: w_prev = w ; s = Pi(w_s) for Zero / Nonnegatives / Box rows, copy for the others
// (solver.jl:151, 14 ; convexset.jl:25-28, 71-74, 844-847 with clip algebra.jl:5-7)
// meta[i] = kind | (boxindex << 2), kind: 0 copy, 1 zero, 2 nonneg, 3 box
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ real project_simple(real x, uint32_t meta, const real* __restrict__ bl,
                                                 const real* __restrict__ bu) {
  const uint32_t kind = meta & 3u;
  if (kind == 0u) return x;
  if (kind == 1u) return 0.0;
  if (kind == 2u) {
    // Julia max(x, 0.0): NaN propagates, max(-0.0, 0.0) == +0.0
    return (x != x) ? x : ((x > R(0.0)) ? x : R(0.0));
  }
  const uint32_t j = meta >> 2;
  const real l = bl[j], u = bu[j];
  return (x < l) ? l : ((x > u) ? u : x);
}

__global__ __launch_bounds__(COSMO_BS) void k_z(const Ctl* __restrict__ ctl, int guard, long long n, long long m,
                                                const real* __restrict__ w, real* __restrict__ w_prev,
                                                real* __restrict__ s, const uint32_t* __restrict__ meta,
                                                const real* __restrict__ bl, const real* __restrict__ bu) {
  if (guard && ctl->halt) return;
  const long long N = n + m;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    const real v = w[i];
    w_prev[i] = v;
    if (i >= n) {
      const long long r = i - n;
      s[r] = project_simple(v, meta[r], bl, bu);
    }
  }
}

// in-place variant for the fine-grained ABI (cosmo_hip_project)
__global__ __launch_bounds__(COSMO_BS) void k_project_simple_inplace(long long m, real* __restrict__ s,
                                                                     const uint32_t* __restrict__ meta,
                                                                     const real* __restrict__ bl,
                                                                     const real* __restrict__ bu) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS)
    s[i] = project_simple(s[i], meta[i], bl, bu);
}

// ---------------------------------------------------------------------------------------------------------------------
// SecondOrderCone projection, one wave per cone, in place on s (convexset.jl:100-114)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_soc(const Ctl* __restrict__ ctl, int guard, int ncones,
                                                  const int* __restrict__ off, const int* __restrict__ dim,
                                                  real* __restrict__ s, int* __restrict__ branch) {
  if (guard && ctl->halt) return;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * COSMO_BS + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * COSMO_BS) >> 6;
  for (int c = wave; c < ncones; c += nwaves) {
    real* x = s + off[c];
    const int d = dim[c];
    if (d == 0) { if (lane == 0) branch[c] = 0; continue; }
    const real t = x[0];
    real acc = 0.0;
    for (int i = 1 + lane; i < d; i += 64) { const real v = x[i]; acc += v * v; }
    const real nx = sqrt(wave_sum(acc));
    int br;
    if (nx <= t) {
      br = 0;
    } else if (nx <= -t) {
      br = 1;
      for (int i = lane; i < d; i += 64) x[i] = 0.0;
    } else {
      br = 2;
      const real f = (nx + t) / (R(2.0) * nx);
      for (int i = 1 + lane; i < d; i += 64) x[i] = f * x[i];
      if (lane == 0) x[0] = (nx + t) / R(2.0);
    }
    if (lane == 0) branch[c] = br;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// warm start: w[1:n] = x0 ; w[n+1:] = 1/rho * mu0 + s0 ; s = s0        (solver.jl:128-129)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_set_w(long long n, long long m, const real* __restrict__ x0,
                                                    const real* __restrict__ s0, const real* __restrict__ mu0,
                                                    const real* __restrict__ rho, real* __restrict__ w,
                                                    real* __restrict__ s) {
  const long long N = n + m;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    if (i < n) {
      w[i] = x0 ? x0[i] : 0.0;
    } else {
      const long long r = i - n;
      const real sv = s0 ? s0[r] : 0.0;
      const real mv = mu0 ? mu0[r] : 0.0;
      w[i] = (R(1.0) / rho[r]) * mv + sv;
      s[r] = sv;
    }
  }
}

// mu = rho .* (w_prev[n+1:] - s)                                          (recover_mu!, solver.jl:24-26)
__global__ __launch_bounds__(COSMO_BS) void k_recover_mu(long long n, long long m, const real* __restrict__ w_prev,
                                                         const real* __restrict__ s, const real* __restrict__ rho,
                                                         real* __restrict__ mu) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS)
    mu[i] = rho[i] * (w_prev[n + i] - s[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// admm_x! right-hand side (solver.jl:50-51) + first line of the reduced solve (kktsolver_indirect.jl:52):
//   ls_x = sigma*w_x - q ; ls_s = (b - 2 s) + w_s ; y2 = rho .* ls_s ; resets the per-solve flags.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_rhs(Ctl* __restrict__ ctl, int guard, long long n, long long m, real sigma,
                                                  const real* __restrict__ w, const real* __restrict__ s,
                                                  const real* __restrict__ q, const real* __restrict__ b,
                                                  const real* __restrict__ rho, real* __restrict__ ls_x,
                                                  real* __restrict__ ls_s, real* __restrict__ y2) {
  if (guard && ctl->halt) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->cg_done = 0; ctl->cg_k = 0; }
  const long long N = n + m;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < N; i += (long long)gridDim.x * COSMO_BS) {
    if (i < n) {
      ls_x[i] = sigma * w[i] - q[i];
    } else {
      const long long r = i - n;
      const real v = (b[r] - R(2.0) * s[r]) + w[i];
      ls_s[r] = v;
      y2[r] = rho[r] * v;
    }
  }
}

// fine-grained solve: y2 = rho .* rhs_s (rhs already uploaded into ls_x / ls_s)
__global__ __launch_bounds__(COSMO_BS) void k_y2_only(Ctl* __restrict__ ctl, long long m, const real* __restrict__ ls_s,
                                                      const real* __restrict__ rho, real* __restrict__ y2) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->cg_done = 0; ctl->cg_k = 0; }
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS)
    y2[i] = rho[i] * ls_s[i];
}

// rhs = A' y2 + ls_x ; partial sum of rhs^2                                (kktsolver_indirect.jl:53-54, 70)
__global__ __launch_bounds__(COSMO_BS) void k_cg_rhs(const Ctl* __restrict__ ctl, int guard, CsrView AT,
                                                     const real* __restrict__ y2, const real* __restrict__ ls_x,
                                                     real* __restrict__ rhs, real* __restrict__ part_bb) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real acc = 0.0;
  for (int k = blockIdx.x; k < AT.nb; k += gridDim.x) {
    csr_stream_tile(AT, y2, y2, k, lds, red, [&](int r, real s1, real s2) {
      const real v = (s1 + s2) + ls_x[r];
      rhs[r] = v;
      acc += v * v;
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_bb[blockIdx.x] = acc;
}

// out = rho .* (A v)                                                       (reduced_mul!, kktsolver_indirect.jl:59-60)
// mode 0: solve start (v = previous solution)   mode 1: Krylov iteration (skipped once the solve has converged)
__global__ __launch_bounds__(COSMO_BS) void k_spmv_A_rho(const Ctl* __restrict__ ctl, int guard, int mode, CsrView A,
                                                         const real* __restrict__ v, const real* __restrict__ rho,
                                                         real* __restrict__ out) {
  if (guard && ctl->halt) return;
  if (mode == 1 && ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  for (int b = blockIdx.x; b < A.nb; b += gridDim.x) {
    csr_stream_tile(A, v, v, tile_of_block(b, A.nb, A.xcd_affine), lds, red,
                     [&](int r, real s1, real s2) { out[r] = (s1 + s2) * rho[r]; });
  }
}

// c = P v + (sigma v + A' tmp) through the row-merged operator [P | A']      (reduced_mul!, :61-64)
// mode 0 (solve start): r = rhs - c, partial sum r^2, block 0 derives the absolute tolerance
//                       tol = tol_k / ||rhs||  (kktsolver_indirect.jl:70 ; cg! abstol)
// mode 1 (iteration)  : store c, partial sum u.c        mode 2: as 1 but ignores the solve flags (timing hook)
__global__ __launch_bounds__(COSMO_BS) void k_op_apply(Ctl* __restrict__ ctl, int guard, int mode, CsrView PT, real sigma,
                                                       const real* __restrict__ v, const real* __restrict__ tmp,
                                                       const real* __restrict__ rhs, real* __restrict__ r,
                                                       real* __restrict__ c, real* __restrict__ part_out,
                                                       const real* __restrict__ part_bb, int n_bb, real tol_k,
                                                       const real* __restrict__ diag) {
  if (guard && ctl->halt) return;
  if (mode == 1 && ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  if (mode == 0 && blockIdx.x == 0) {
    const real bb = reduce_partials_sum(part_bb, n_bb, red);
    if (threadIdx.x == 0) {
      const real nb = sqrt(bb);
      ctl->rhs_norm = nb;
      ctl->tol = tol_k / nb;
    }
  }
  real acc = 0.0;
  const int first_tile = tile_of_block(blockIdx.x, PT.nb, PT.xcd_affine);     // affine => grid == nb: the loop runs once
  for (int k = first_tile; k < PT.nb; k += gridDim.x) {
    csr_stream_tile(PT, v, tmp, k, lds, red, [&](int row, real s1, real s2) {
      const real vj = v[row];
      real cj = s1 + (sigma * vj + s2);
      if (diag) cj += diag[row] * vj;      // singleton rows of A: their part of A' rho A is diagonal (build_op_split)
      if (mode == 0) {
        const real rj = rhs[row] - cj;
        r[row] = rj;
        acc += rj * rj;
      } else {
        c[row] = cj;
        acc += vj * cj;
      }
    });
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_out[PT.xcd_affine ? first_tile : (int)blockIdx.x] = acc;      // indexed by TILE: independent of the tile order
}

// Krylov step k, first half (IterativeSolvers v0.9 cg.jl `iterate`): residual_k = ||r|| from the partials;
// stop if k >= maxiter or residual_k <= tol (checked BEFORE the iteration); else beta = res_k^2 / res_{k-1}^2,
// u = r + beta u (u_{-1} = 0).  check_only = 1: evaluate the stopping rule after the last budgeted iteration.
__global__ __launch_bounds__(COSMO_BS) void k_cg_dir(Ctl* __restrict__ ctl, int guard, int k, int check_only, long long n,
                                                     long long maxiter, const real* __restrict__ part_rr, int n_rr,
                                                     const real* __restrict__ r, real* __restrict__ u) {
  // operands of the elementwise part are requested BEFORE the scalar work so that their latency overlaps the
  // partial reduction (the grid covers n with one element per thread; the strided loop only handles huge n)
  const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
  real r0 = 0.0, u0 = 0.0;
  if (!check_only && i0 < n) { r0 = r[i0]; if (k != 0) u0 = u[i0]; }
  const real pa = partials_prefetch_sum(part_rr, n_rr);      // in flight while the guards wait on their scalar loads
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real red[COSMO_BS / 64];
  if (k < 0) k = ctl->cg_k;                      // device-side index (the check behind a captured chain, cg_fold.hip): nobody writes cg_k during this kernel
  const real tol = ctl->tol;
  const real prev = (k == 0) ? 1.0 : ctl->resv[(k - 1) & 1];
  const real rr = block_sum(pa, red);
  const real res = sqrt(rr);
  const bool done = (k >= maxiter) || (res <= tol);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) ctl->cg_done = 1;
    if (!check_only || done) ctl->resv[k & 1] = res;
  }
  if (done || check_only) return;
  const real beta = (res * res) / (prev * prev);
  if (i0 < n) u[i0] = r0 + beta * u0;
  for (long long i = i0 + (long long)gridDim.x * COSMO_BS; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const real ui = (k == 0) ? 0.0 : u[i];
    u[i] = r[i] + beta * ui;
  }
}

// Krylov step k >= 1, first half FUSED with the A product (one launch and one kernel boundary less per iteration): every workgroup
// folds the r'r partials itself (as k_cg_dir does), applies the stopping rule, and computes tmp = rho .* (A u_k) with u_k = r + beta
// u_{k-1} REBUILT at the gathered columns -- same expression, same bits as the owner's update.  r and u_{k-1} live interleaved in
// `ru` ({r_i, u_i}, written by k_cg_upd), so the two operands of a column arrive with ONE 16-byte gather.  The workgroups also
// materialise u_k (grid-stride over the elements) for the operator kernel and for k_cg_upd.
__global__ __launch_bounds__(COSMO_BS) void k_cg_dirA(Ctl* __restrict__ ctl, int guard, int check_first, int k, long long n, long long maxiter,
                                                      const real* __restrict__ part_rr, int n_rr, CsrView A, const real2* __restrict__ ru,
                                                      const real* __restrict__ rho, real* __restrict__ tmp, real* __restrict__ u) {
  // check_first: an iteration the host expects to lie BEHIND the end of the solve (k beyond the largest recent Krylov count, api.hip:
  // solve_budget) looks at the flags before it requests anything -- a no-op launch then costs ~1.5 us instead of the 11 us its gathers
  // take to drain; if the solve does reach it, it pays the flags' round trip once.
  if (check_first) { if (guard && ctl->halt) return; if (ctl->cg_done) return; }
  // Everything that does not depend on beta is requested FIRST -- the r'r partials, the tile descriptor, (col, val) and the 16-byte
  // {r, u} gathers of the first tile, the row pointers, this workgroup's slice of {r, u} -- so that the partial reduction / stopping
  // rule overlaps the gather latency instead of preceding it.  Absent slots read a valid address and carry a zero matrix value
  // (a load under a per-thread `if` would make the compiler wait for it on the spot).
  constexpr int SL = COSMO_NNZ_PER_BLOCK / COSMO_BS;
  const real pa = partials_prefetch_sum(part_rr, n_rr);
  const bool have_tile = (int)blockIdx.x < A.nb;
  int4 d = make_int4(0, 0, 0, 0);
  if (have_tile) d = reinterpret_cast<const int4*>(A.rb)[tile_of_block(blockIdx.x, A.nb, A.xcd_affine)];
  const int cnt0 = d.w - d.z;
  const bool fast = have_tile && cnt0 <= COSMO_NNZ_PER_BLOCK;      // a single long row takes the generic chunked path below
  real av[SL]; real2 gv[SL];
#pragma unroll
  for (int it = 0; it < SL; ++it) {
    const int kk = it * COSMO_BS + threadIdx.x;
    const bool ok = fast && kk < cnt0;
    const int e = ok ? d.z + kk : 0;
    const int c = A.col[e];
    const real a = A.val[e];
    av[it] = ok ? a : 0.0;
    gv[it] = ru[c];
  }
  const int rfirst = d.x + threadIdx.x;
  const bool rowok = fast && rfirst < d.y;
  const int rr_ = rowok ? rfirst : 0;
  const int pa_ = A.rowptr[rr_], pb_ = A.rowptr[rr_ + 1];
  const real rho_ = rho[rr_];
  const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
  const real2 own = ru[i0 < n ? i0 : 0];
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  const real tol = ctl->tol;
  const real prev = ctl->resv[(k - 1) & 1];
  const real rr = block_sum(pa, red);
  const real res = sqrt(rr);
  const bool done = (k >= maxiter) || (res <= tol);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) ctl->cg_done = 1;
    ctl->resv[k & 1] = res;
  }
  if (done) return;
  const real beta = (res * res) / (prev * prev);
  if (i0 < n) u[i0] = own.x + beta * own.y;
  for (long long i = i0 + (long long)gridDim.x * COSMO_BS; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const real2 v = ru[i];
    u[i] = v.x + beta * v.y;
  }
  if (fast) {
#pragma unroll
    for (int it = 0; it < SL; ++it) {
      const int kk = it * COSMO_BS + threadIdx.x;
      if (kk < cnt0) lds[kk] = av[it] * (gv[it].x + beta * gv[it].y);
    }
    __syncthreads();
    if (rowok) {                                    // first row of this thread: pointers already here
      const real s1 = lds_seq_sum(lds, pa_ - d.z, pb_ - d.z), s2 = 0.0;
      tmp[rfirst] = (s1 + s2) * rho_;
    }
    for (int r = rfirst + COSMO_BS; r < d.y; r += COSMO_BS) {
      const real s1 = lds_seq_sum(lds, A.rowptr[r] - d.z, A.rowptr[r + 1] - d.z), s2 = 0.0;
      tmp[r] = (s1 + s2) * rho[r];
    }
    __syncthreads();
  }
  for (int b = fast ? (int)(blockIdx.x + gridDim.x) : (int)blockIdx.x; b < A.nb; b += gridDim.x) {
    const int4 dd = reinterpret_cast<const int4*>(A.rb)[tile_of_block(b, A.nb, A.xcd_affine)];
    csr_stream_rows_g(A, [&](int c) { const real2 v = ru[c]; return v.x + beta * v.y; }, dd.x, dd.y, dd.z, dd.w, lds, red,
                      [&](int r, real s1, real s2) { tmp[r] = (s1 + s2) * rho[r]; });
  }
}

// Krylov step k, second half: alpha = res_k^2 / (u.c) ; x += alpha u ; r -= alpha c ; partial sum r^2
// PC (opt-in Jacobi PCG on the assembled operator, cg_fold.hip): alpha = rho_k / (u.c) with rho_k = z'r (ctl->sr_gamma), the records become
// {z_{k+1}, u_k} with z = dinv .* r, and the partials of z'r go to part_rz.
template <bool PC>
__global__ __launch_bounds__(COSMO_BS) void k_cg_upd(Ctl* __restrict__ ctl, int guard, int k, long long n,
                                                     const real* __restrict__ part_uc, int n_uc,
                                                     const real* __restrict__ u, const real* __restrict__ c,
                                                     real* __restrict__ x, real* __restrict__ r,
                                                     real* __restrict__ part_rr, real2* __restrict__ ru,
                                                     const real* __restrict__ dinv, real* __restrict__ part_rz) {
  const long long i0 = (long long)blockIdx.x * COSMO_BS + threadIdx.x;
  real u0 = 0.0, c0 = 0.0, x0 = 0.0, r0 = 0.0, d0 = 0.0;
  if (i0 < n) { u0 = u[i0]; c0 = c[i0]; x0 = x[i0]; r0 = r[i0]; if constexpr (PC) d0 = dinv[i0]; }   // issued before the scalar work (latency overlap)
  const real pa = partials_prefetch_sum(part_uc, n_uc);
  if (guard && ctl->halt) return;
  if (ctl->cg_done) return;
  __shared__ real red[COSMO_BS / 64];
  if (k < 0) k = ctl->cg_kd;                     // device-side index: published by the k_cg_dirM in front (cg_k itself is rewritten below by workgroup 0)
  const real res = ctl->resv[k & 1];
  const real uc = block_sum(pa, red);
  real alpha = (res * res) / uc;
  if constexpr (PC) alpha = ctl->sr_gamma[k & 1] / uc;
  real acc = 0.0, accz = 0.0;
  if (i0 < n) {
    x[i0] = x0 + alpha * u0;
    const real ri = r0 - alpha * c0;
    r[i0] = ri;
    if constexpr (PC) {
      const real zi = d0 * ri;
      ru[i0] = make_real2(zi, u0);
      accz += zi * ri;
    } else {
      if (ru) ru[i0] = make_real2(ri, u0);     // {r_{k+1}, u_k}: the operands of the fused direction + product kernel
    }
    acc += ri * ri;
  }
  for (long long i = i0 + (long long)gridDim.x * COSMO_BS; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const real ui = u[i];
    x[i] = x[i] + alpha * ui;
    const real ri = r[i] - alpha * c[i];
    r[i] = ri;
    if constexpr (PC) {
      const real zi = dinv[i] * ri;
      ru[i] = make_real2(zi, ui);
      accz += zi * ri;
    } else {
      if (ru) ru[i] = make_real2(ri, ui);
    }
    acc += ri * ri;
  }
  acc = block_sum(acc, red);
  if constexpr (PC) accz = block_sum(accz, red);
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x] = acc;
    if constexpr (PC) part_rz[blockIdx.x] = accz;
    if (blockIdx.x == 0) ctl->cg_k = k + 1;
  }
}

// End of solve! + rest of admm_x! + admm_w!:
//   nu = rho .* (A x_tl - ls_s)                     (kktsolver_indirect.jl:81-83)
//   s_tl = (2 s - w_s) - nu ./ rho                  (solver.jl:55)
//   w_s = w_s + alpha (s_tl - s) ; w_x = w_x + alpha (x_tl - w_x)      (solver.jl:63-64)
// loop_mode 1: also detects an exhausted Krylov budget (stall) and advances the device counters.
// loop_mode 0: fine-grained solve (only nu is produced).
__global__ __launch_bounds__(COSMO_BS) void k_tail(Ctl* __restrict__ ctl, int loop_mode, CsrView A, long long n, long long m,
                                                   real alpha, const real* __restrict__ x_tl,
                                                   const real* __restrict__ ls_s, const real* __restrict__ rho,
                                                   const real* __restrict__ s, real* __restrict__ nu,
                                                   real* __restrict__ s_tl, real* __restrict__ w, int nblk_rows) {
  if (loop_mode) {
    if (ctl->halt) return;
    if (!ctl->cg_done) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->stalled = 1; ctl->halt = 1; }
      return;
    }
  }
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  if ((int)blockIdx.x < nblk_rows) {
    for (int k = blockIdx.x; k < A.nb; k += nblk_rows) {
      csr_stream_tile(A, x_tl, x_tl, k, lds, red, [&](int row, real s1, real s2) {
        const real ax = s1 + s2;
        const real rh = rho[row];
        const real nv = (ax - ls_s[row]) * rh;
        nu[row] = nv;
        if (loop_mode) {
          const real sv = s[row];
          const real wv = w[n + row];
          const real st = (R(2.0) * sv - wv) - nv / rh;
          s_tl[row] = st;
          w[n + row] = wv + alpha * (st - sv);
        }
      });
    }
  } else if (loop_mode) {
    const long long nb2 = gridDim.x - nblk_rows;
    for (long long i = (long long)(blockIdx.x - nblk_rows) * COSMO_BS + threadIdx.x; i < n; i += nb2 * COSMO_BS) {
      const real wv = w[i];
      w[i] = wv + alpha * (x_tl[i] - wv);
    }
    if ((int)blockIdx.x == nblk_rows && threadIdx.x == 0) {
      const int k = ctl->cg_k;
      ctl->iter += 1;
      ctl->solves += 1;
      ctl->kkt_iters_total += k;
      if (k > ctl->cg_k_max) ctl->cg_k_max = k;
    }
  }
}

// counters for the fine-grained solve
__global__ void k_count_solve(Ctl* __restrict__ ctl) {
  const int k = ctl->cg_k;
  ctl->solves += 1;
  ctl->kkt_iters_total += k;
  if (k > ctl->cg_k_max) ctl->cg_k_max = k;
}

// ---------------------------------------------------------------------------------------------------------------------
// residual checks (residuals.jl:30-96, 143-147).  x = w_prev[1:n], mu = rho .* (w_prev_s - s) recovered on the fly.
// primal pass over A:  r_prim = A x + s - b ; norms of Einv-scaled r_prim, A x, s, b
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_chk_prim(const Ctl* __restrict__ ctl, int guard, CsrView A, long long n,
                                                       const real* __restrict__ w_prev, const real* __restrict__ s,
                                                       const real* __restrict__ b, const real* __restrict__ rho,
                                                       const real* __restrict__ Einv, real* __restrict__ mu,
                                                       real* __restrict__ part_rp, real* __restrict__ part_mp) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real rp = 0.0, mp = 0.0;
  for (int k = blockIdx.x; k < A.nb; k += gridDim.x) {
    csr_stream_tile(A, w_prev, w_prev, k, lds, red, [&](int row, real s1, real s2) {
      const real ax = s1 + s2;
      const real sv = s[row];
      const real bv = b[row];
      mu[row] = rho[row] * (w_prev[n + row] - sv);
      real rv = ax + sv;
      rv = rv - bv;
      real e = 1.0;
      if (Einv) e = Einv[row];
      if (Einv) rv = rv * e;
      rp = amax(rp, rv);
      mp = amax(mp, Einv ? ax * e : ax);
      mp = amax(mp, Einv ? sv * e : sv);
      mp = amax(mp, Einv ? bv * e : bv);
    });
  }
  rp = block_max(rp, red);
  mp = block_max(mp, red);
  if (threadIdx.x == 0) { part_rp[blockIdx.x] = rp; part_mp[blockIdx.x] = mp; }
}

// dual pass over [P | A']:  r_dual = P x + q - A' mu ; norms of cinv*Dinv-scaled r_dual, P x, q, A' mu ; x'Px, q'x
__global__ __launch_bounds__(COSMO_BS) void k_chk_dual(const Ctl* __restrict__ ctl, int guard, CsrView PT,
                                                       const real* __restrict__ w_prev, const real* __restrict__ mu,
                                                       const real* __restrict__ q, const real* __restrict__ Dinv,
                                                       real cinv, int unscale, real* __restrict__ part_rd,
                                                       real* __restrict__ part_md, real* __restrict__ part_xpx,
                                                       real* __restrict__ part_qx) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real rd = 0.0, md = 0.0, xpx = 0.0, qx = 0.0;
  for (int k = blockIdx.x; k < PT.nb; k += gridDim.x) {
    csr_stream_tile(PT, w_prev, mu, k, lds, red, [&](int row, real px, real atm) {
      const real xv = w_prev[row];
      const real qv = q[row];
      real rv = px + qv;
      rv = rv - atm;
      real a = px, bq = qv, cm = atm;
      if (unscale) {
        const real d = Dinv ? Dinv[row] : 1.0;
        rv = (rv * d) * cinv;
        a = (a * d) * cinv;
        bq = (bq * d) * cinv;
        cm = (cm * d) * cinv;
      }
      rd = amax(rd, rv);
      md = amax(md, a);
      md = amax(md, bq);
      md = amax(md, cm);
      xpx += px * xv;
      qx += qv * xv;
    });
  }
  rd = block_max(rd, red);
  md = block_max(md, red);
  xpx = block_sum(xpx, red);
  qx = block_sum(qx, red);
  if (threadIdx.x == 0) {
    part_rd[blockIdx.x] = rd; part_md[blockIdx.x] = md; part_xpx[blockIdx.x] = xpx; part_qx[blockIdx.x] = qx;
  }
}

// One workgroup: fold the partials, then take the data-dependent decision on the device.
// mode 0 = info only (calculate_result_info!, residuals.jl:149-153)
// mode 1 = check_termination! (solver.jl:306-321): cost, Unsolved / Solved
// mode 3 = info + cost without a decision (cosmo_hip_residuals)
// mode 2 = adapt_rho_vec! (parameters.jl:53-72): scalar rule; sets rho_changed for k_rho_apply
struct ChkArgs {
  const real *part_rp, *part_mp, *part_rd, *part_md, *part_xpx, *part_qx;
  int n_prim, n_dual;
  int mode;
  real cinv, eps_abs, eps_rel;
  real obj_true, obj_true_tol;
  real rho_min, rho_max, adapt_tol;
  long long max_adaptions;
};
__global__ __launch_bounds__(COSMO_BS) void k_chk_final(Ctl* __restrict__ ctl, int guard, ChkArgs a) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  const real rp = reduce_partials_max(a.part_rp, a.n_prim, red);
  const real mp = reduce_partials_max(a.part_mp, a.n_prim, red);
  const real rd = reduce_partials_max(a.part_rd, a.n_dual, red);
  const real md = reduce_partials_max(a.part_md, a.n_dual, red);
  const real xpx = reduce_partials_sum(a.part_xpx, a.n_dual, red);
  const real qx = reduce_partials_sum(a.part_qx, a.n_dual, red);
  if (threadIdx.x != 0) return;
  if (a.mode == 2) {
    ctl->rho_changed = 0;
    if ((long long)(ctl->n_rho_updates - 1) >= a.max_adaptions) return;
    const real rpn = rp / (mp + R(1e-10));
    const real rdn = rd / (md + R(1e-10));
    const real rho = ctl->rho;
    real nr = rho * sqrt(rpn / (rdn + R(1e-10)));
    nr = clamp_keep_nan(nr, a.rho_min, a.rho_max);
    if ((nr > a.adapt_tol * rho) || (nr < (R(1.0) / a.adapt_tol) * rho)) {
      ctl->rho = nr;
      ctl->rho_changed = 1;
      const int k = ctl->n_rho_updates;
      if (k < COSMO_HIP_MAX_RHO_UPDATES) ctl->rho_updates[k] = nr;
      ctl->n_rho_updates = k + 1;
    }
    return;
  }
  ctl->r_prim = rp; ctl->r_dual = rd; ctl->max_norm_prim = mp; ctl->max_norm_dual = md;
  if (a.mode == 1 || a.mode == 3) {
    const real cost = a.cinv * (R(0.5) * xpx + qx);
    ctl->cost = cost;
    if (a.mode == 3) return;
    if (fabs(cost) > R(1e20)) { ctl->status = COSMO_HIP_UNSOLVED; ctl->halt = 1; return; }
    const bool pf = rp < a.eps_abs + a.eps_rel * mp;
    const bool df = rd < a.eps_abs + a.eps_rel * md;
    const bool ot = (a.obj_true != a.obj_true) || (fabs(a.obj_true - cost) <= a.obj_true_tol);   // has_converged (residuals.jl:131-139)
    if (pf && df && ot) { ctl->status = COSMO_HIP_SOLVED; ctl->halt = 1; }
  }
}

// update_rho_vec! (parameters.jl:75-92) + w_s = 1/rho * mu + s (solver.jl:278), only if the rule fired
__global__ __launch_bounds__(COSMO_BS) void k_rho_apply(const Ctl* __restrict__ ctl, int guard, long long n, long long m,
                                                        const int* __restrict__ cls, real rho_min, real rho_eq,
                                                        const real* __restrict__ mu, const real* __restrict__ s,
                                                        real* __restrict__ rho, real* __restrict__ w) {
  if (guard && ctl->halt) return;
  if (!ctl->rho_changed) return;
  const real nr = ctl->rho;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS) {
    const int c = cls[i];
    real rv = nr;
    if (c == 1) rv = rv * rho_eq; else if (c == 2) rv = rho_min;
    rho[i] = rv;
    w[n + i] = (R(1.0) / rv) * mu[i] + s[i];
  }
}

// rho vector from classes (set_rho_vec!, parameters.jl:3-13)
__global__ __launch_bounds__(COSMO_BS) void k_rho_from_classes(long long m, const int* __restrict__ cls, real rho0,
                                                               real rho_min, real rho_eq, real* __restrict__ rho) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m; i += (long long)gridDim.x * COSMO_BS) {
    const int c = cls[i];
    real rv = rho0;
    if (c == 1) rv = rv * rho_eq; else if (c == 2) rv = rho_min;
    rho[i] = rv;
  }
}

__global__ void k_ctl_clear_stall(Ctl* ctl) { ctl->stalled = 0; ctl->halt = (ctl->status != 0 || ctl->error != 0) ? 1 : 0; }
__global__ void k_ctl_set_done(Ctl* ctl) { ctl->cg_done = 1; }

// ---------------------------------------------------------------------------------------------------------------------
// Row-sharded runs (rowshard.hip; SURVEY 8e option 2 on a replicated CG).  Rank g owns the rows of its cones: A_g = rows of A (h->A),
// A_g' = columns of A' (h->AT), slices of every m-vector.  A' y = sum_g A_g' y_g is the ONE quantity of the iteration that needs the
// other ranks: the partial products are summed by an all-reduce (comm.hip) and these kernels finish what k_cg_rhs / k_chk_dual do in
// one pass on a single GPU.
// ---------------------------------------------------------------------------------------------------------------------
// rhs = (sum over ranks of A_g' y2_g) + ls_x ; partial sums of rhs^2               (kktsolver_indirect.jl:53-54, 70)
__global__ __launch_bounds__(COSMO_BS) void k_rs_rhs_fin(const Ctl* __restrict__ ctl, int guard, long long n, const real* __restrict__ aty,
                                                         const real* __restrict__ ls_x, real* __restrict__ rhs, real* __restrict__ part_bb) {
  if (guard && ctl->halt) return;
  __shared__ real red[COSMO_BS / 64];
  real acc = 0.0;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n; i += (long long)gridDim.x * COSMO_BS) {
    const real v = aty[i] + ls_x[i];
    rhs[i] = v;
    acc += v * v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part_bb[blockIdx.x] = acc;
}

// this rank's primal norms (max over its partials) into its two slots behind the n-vector that is about to be all-reduced; the other
// ranks' slots are zeroed, so the SUM over the ranks delivers every rank's value to everybody (NaN included) with the same collective
__global__ __launch_bounds__(COSMO_BS) void k_rs_pack_prim(const real* __restrict__ part_rp, const real* __restrict__ part_mp, int n_parts,
                                                           real* __restrict__ out, int rank, int nranks) {
  __shared__ real red[COSMO_BS / 64];
  const real rp = reduce_partials_max(part_rp, n_parts, red);
  const real mp = reduce_partials_max(part_mp, n_parts, red);
  if ((int)threadIdx.x < 2 * nranks) out[threadIdx.x] = R(0.0);
  __syncthreads();
  if (threadIdx.x == 0) { out[rank] = rp; out[nranks + rank] = mp; }
}

// dual pass with A' mu already summed over the ranks: r_dual = P x + q - atm ; norms ; x'Px, q'x       (residuals.jl:12-18, 56-96, 143-147)
__global__ __launch_bounds__(COSMO_BS) void k_chk_dual_rs(const Ctl* __restrict__ ctl, int guard, CsrView P, const real* __restrict__ w_prev,
                                                          const real* __restrict__ atmv, const real* __restrict__ q,
                                                          const real* __restrict__ Dinv, real cinv, int unscale, real* __restrict__ part_rd,
                                                          real* __restrict__ part_md, real* __restrict__ part_xpx, real* __restrict__ part_qx) {
  if (guard && ctl->halt) return;
  __shared__ real lds[COSMO_NNZ_PER_BLOCK];
  __shared__ real red[COSMO_BS / 64];
  real rd = 0.0, md = 0.0, xpx = 0.0, qx = 0.0;
  for (int k = blockIdx.x; k < P.nb; k += gridDim.x) {
    csr_stream_tile(P, w_prev, w_prev, k, lds, red, [&](int row, real s1, real s2) {
      const real px = s1 + s2;
      const real atm = atmv[row];
      const real xv = w_prev[row];
      const real qv = q[row];
      real rv = px + qv;
      rv = rv - atm;
      real a = px, bq = qv, cm = atm;
      if (unscale) {
        const real d = Dinv ? Dinv[row] : 1.0;
        rv = (rv * d) * cinv;
        a = (a * d) * cinv;
        bq = (bq * d) * cinv;
        cm = (cm * d) * cinv;
      }
      rd = amax(rd, rv);
      md = amax(md, a);
      md = amax(md, bq);
      md = amax(md, cm);
      xpx += px * xv;
      qx += qv * xv;
    });
  }
  rd = block_max(rd, red);
  md = block_max(md, red);
  xpx = block_sum(xpx, red);
  qx = block_sum(qx, red);
  if (threadIdx.x == 0) {
    part_rd[blockIdx.x] = rd; part_md[blockIdx.x] = md; part_xpx[blockIdx.x] = xpx; part_qx[blockIdx.x] = qx;
  }
}

// update_rho_vec! on the replicated full-length rho that the reduced operator is refreshed from (every rank, identical)
__global__ __launch_bounds__(COSMO_BS) void k_rs_rho_g(const Ctl* __restrict__ ctl, int guard, long long m_g, const int* __restrict__ cls,
                                                       real rho_min, real rho_eq, real* __restrict__ rho_g) {
  if (guard && ctl->halt) return;
  if (!ctl->rho_changed) return;
  const real nr = ctl->rho;
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < m_g; i += (long long)gridDim.x * COSMO_BS) {
    const int c = cls[i];
    real rv = nr;
    if (c == 1) rv = rv * rho_eq; else if (c == 2) rv = rho_min;
    rho_g[i] = rv;
  }
}

// =====================================================================================================================
// host-side launchers
// =====================================================================================================================
static inline int ew_grid(long long N) {
  long long g = (N + COSMO_BS - 1) / COSMO_BS;
  if (g < 1) g = 1;
  if (g > COSMO_MAX_PARTIALS) g = COSMO_MAX_PARTIALS;
  return (int)g;
}

#define PARTS(h, slot) ((h)->partials + (size_t)(slot) * COSMO_MAX_PARTIALS)

int32_t launch_spmv_plain(cosmo_hip_handle* h, const CsrDev& M, const real* x, real* y) {
  if (M.nrows == 0) return COSMO_HIP_OK;
  hipLaunchKernelGGL(k_spmv_plain, dim3(M.grid), dim3(COSMO_BS), 0, h->stream, view_of(M), x, y);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_project_simple_inplace(cosmo_hip_handle* h, real* s) {
  if (h->m == 0) return COSMO_HIP_OK;
  hipLaunchKernelGGL(k_project_simple_inplace, dim3(ew_grid(h->m)), dim3(COSMO_BS), 0, h->stream, h->m, s, h->meta,
                     h->box_l, h->box_u);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_z(cosmo_hip_handle* h, int guard) {
  prof_begin(h, KC_Z);
  hipLaunchKernelGGL(k_z, dim3(ew_grid(h->n + h->m)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->n, h->m, h->w,
                     h->w_prev, h->s, h->meta, h->box_l, h->box_u);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_soc(cosmo_hip_handle* h, real* s, int guard) {
  if (h->nsoc == 0) return COSMO_HIP_OK;
  int g = (h->nsoc + (COSMO_BS / 64) - 1) / (COSMO_BS / 64);
  if (g > 4096) g = 4096;
  prof_begin(h, KC_SOC);
  hipLaunchKernelGGL(k_soc, dim3(g), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->nsoc, h->soc_off, h->soc_dim, s,
                     h->soc_branch);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_set_w(cosmo_hip_handle* h, const real* x0, const real* s0, const real* mu0) {
  hipLaunchKernelGGL(k_set_w, dim3(ew_grid(h->n + h->m)), dim3(COSMO_BS), 0, h->stream, h->n, h->m, x0, s0, mu0, h->rho,
                     h->w, h->s);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_recover_mu(cosmo_hip_handle* h) {
  if (h->m == 0) return COSMO_HIP_OK;
  hipLaunchKernelGGL(k_recover_mu, dim3(ew_grid(h->m)), dim3(COSMO_BS), 0, h->stream, h->n, h->m, h->w_prev, h->s, h->rho,
                     h->mu);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t launch_rho_from_classes(cosmo_hip_handle* h, real rho0) {
  if (h->m == 0) return COSMO_HIP_OK;
  hipLaunchKernelGGL(k_rho_from_classes, dim3(ew_grid(h->m)), dim3(COSMO_BS), 0, h->stream, h->m, h->rho_cls, rho0,
                     h->prm.rho_min, h->prm.rho_eq_over_rho_ineq, h->rho);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// ---- reduced-system CG solve, enqueued without host synchronisation --------------------------------------------------
// from_loop: rhs comes from the loop state (k_rhs) and the tail updates w; otherwise ls_x/ls_s were uploaded.
int32_t enqueue_cg_iterations(cosmo_hip_handle* h, int guard, int k_begin, int count) {
  if (h->op_fold) return fold_enqueue_iterations(h, guard, k_begin, count);      // assembled operator: two launches per iteration
  const long long n = h->n;
  const int gE = ew_grid(n);
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  const real* rho_o = h->op_split ? h->op_rho_m : h->rho;
  const real* diag_o = h->op_split ? h->op_diag : nullptr;
  const int n_rr0 = PTo.grid;
  for (int k = k_begin; k < k_begin + count; ++k) {
    if (h->cg_ru && k > 0) {
      prof_begin(h, KC_SPMV_A);
      hipLaunchKernelGGL(k_cg_dirA, dim3(Ao.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, (k >= h->cg_k_likely) ? 1 : 0, k, n, n, PARTS(h, SLOT_RR), gE, view_of(Ao),
                         (const real2*)h->cg_ru, rho_o, h->tmp_m, h->u);
      prof_end(h);
    } else {
      prof_begin(h, KC_CG_DIR);
      hipLaunchKernelGGL(k_cg_dir, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, 0, n, n, PARTS(h, SLOT_RR),
                         (k == 0) ? n_rr0 : gE, h->r, h->u);
      prof_end(h);
      prof_begin(h, KC_SPMV_A);
      hipLaunchKernelGGL(k_spmv_A_rho, dim3(Ao.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 1, view_of(Ao), h->u,
                         rho_o, h->tmp_m);
      prof_end(h);
    }
    prof_begin(h, KC_OP_APPLY);
    hipLaunchKernelGGL(k_op_apply, dim3(PTo.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 1, view_of(PTo),
                       h->prm.sigma, h->u, h->tmp_m, h->rhs, h->r, h->c, PARTS(h, SLOT_UC), PARTS(h, SLOT_BB), 0, 0.0, diag_o);
    prof_end(h);
    prof_begin(h, KC_CG_UPD);
    hipLaunchKernelGGL(k_cg_upd<false>, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, n, PARTS(h, SLOT_UC),
                       PTo.grid, h->u, h->c, h->x_tl, h->r, PARTS(h, SLOT_RR), (real2*)h->cg_ru, (const real*)nullptr, (real*)nullptr);
    prof_end(h);
    h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  }
  // evaluate the stopping rule once more after the last budgeted iteration
  const int kk = k_begin + count;
  prof_begin(h, KC_CG_DIR);
  hipLaunchKernelGGL(k_cg_dir, dim3(1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, kk, 1, n, n, PARTS(h, SLOT_RR),
                     (kk == 0) ? n_rr0 : gE, h->r, h->u);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// row-sharded: rhs = allreduce_sum(A_g' y2_g) + ls_x
int32_t rs_enqueue_cg_rhs(cosmo_hip_handle* h, int guard) {
  const int gE = ew_grid(h->n);
  prof_begin(h, KC_SPMV_AT);
  CHK(launch_spmv_plain(h, h->AT, h->y2, h->red_n));
  prof_end(h);
  CHK(comm_allreduce_sum(h, h->red_n, (size_t)h->n));
  h->rs_allreduces += 1; h->rs_allreduce_elems = h->n;
  prof_begin(h, KC_RHS);
  hipLaunchKernelGGL(k_rs_rhs_fin, dim3(gE), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->n, h->red_n, h->ls_x, h->rhs, PARTS(h, SLOT_BB));
  prof_end(h);
  h->n_bb = gE;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_cg_start(cosmo_hip_handle* h, int guard, real tol_k) {
  if (h->row_shard) {
    CHK(rs_enqueue_cg_rhs(h, guard));
  } else {
    prof_begin(h, KC_SPMV_AT);
    hipLaunchKernelGGL(k_cg_rhs, dim3(h->AT.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(h->AT), h->y2,
                       h->ls_x, h->rhs, PARTS(h, SLOT_BB));
    prof_end(h);
    h->n_bb = h->AT.grid;
  }
  if (h->op_fold) return fold_enqueue_start(h, guard, tol_k);
  const CsrDev& Ao = h->op_split ? h->Am : h->A;
  const CsrDev& PTo = h->op_split ? h->PTm : h->PT;
  prof_begin(h, KC_SPMV_A);
  hipLaunchKernelGGL(k_spmv_A_rho, dim3(Ao.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 0, view_of(Ao), h->x_tl,
                     h->op_split ? h->op_rho_m : h->rho, h->tmp_m);
  prof_end(h);
  prof_begin(h, KC_OP_APPLY);
  hipLaunchKernelGGL(k_op_apply, dim3(PTo.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, 0, view_of(PTo),
                     h->prm.sigma, h->x_tl, h->tmp_m, h->rhs, h->r, h->c, PARTS(h, SLOT_RR), PARTS(h, SLOT_BB),
                     h->n_bb, tol_k, h->op_split ? h->op_diag : nullptr);
  prof_end(h);
  h->spmv_calls[0] += 1; h->spmv_calls[1] += 2; h->spmv_calls[2] += 1;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_rhs(cosmo_hip_handle* h, int guard) {
  prof_begin(h, KC_RHS);
  hipLaunchKernelGGL(k_rhs, dim3(ew_grid(h->n + h->m)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->n, h->m,
                     h->prm.sigma, h->w, h->s, h->q, h->b, h->rho, h->ls_x, h->ls_s, h->y2);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_y2_only(cosmo_hip_handle* h) {
  hipLaunchKernelGGL(k_y2_only, dim3(ew_grid(h->m > 0 ? h->m : 1)), dim3(COSMO_BS), 0, h->stream, h->ctl, h->m, h->ls_s,
                     h->rho, h->y2);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_tail(cosmo_hip_handle* h, int loop_mode) {
  const int gr = h->A.grid > 0 ? h->A.grid : 1;
  const int gx = loop_mode ? ew_grid(h->n) : 0;
  prof_begin(h, KC_TAIL);
  hipLaunchKernelGGL(k_tail, dim3(gr + gx), dim3(COSMO_BS), 0, h->stream, h->ctl, loop_mode, view_of(h->A), h->n, h->m,
                     h->prm.alpha, h->x_tl, h->ls_s, h->rho, h->s, h->nu, h->s_tl, h->w, gr);
  prof_end(h);
  h->spmv_calls[0] += 1;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_count_solve(cosmo_hip_handle* h) {
  hipLaunchKernelGGL(k_count_solve, dim3(1), dim3(1), 0, h->stream, h->ctl);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t enqueue_clear_stall(cosmo_hip_handle* h) {
  hipLaunchKernelGGL(k_ctl_clear_stall, dim3(1), dim3(1), 0, h->stream, h->ctl);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// mode: 0 info, 1 termination, 2 adaptation (unscaled residuals, parameters.jl:57-59)
int32_t enqueue_check(cosmo_hip_handle* h, int guard, int mode) {
  const bool unscale = (mode != 2) && h->prm.unscale_residuals && h->has_scaling;
  const int nr = h->row_shard ? comm_nranks(h) : 1;
  prof_begin(h, KC_CHK_PRIM);
  hipLaunchKernelGGL(k_chk_prim, dim3(h->A.grid > 0 ? h->A.grid : 1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard,
                     view_of(h->A), h->n, h->w_prev, h->s, h->b, h->rho, unscale ? h->Einv : (const real*)nullptr, h->mu,
                     PARTS(h, SLOT_RP), PARTS(h, SLOT_MP));
  prof_end(h);
  ChkArgs a;
  a.part_rp = PARTS(h, SLOT_RP); a.part_mp = PARTS(h, SLOT_MP); a.part_rd = PARTS(h, SLOT_RD);
  a.part_md = PARTS(h, SLOT_MD); a.part_xpx = PARTS(h, SLOT_XPX); a.part_qx = PARTS(h, SLOT_QX);
  a.n_prim = h->A.grid > 0 ? h->A.grid : 1; a.n_dual = h->PT.grid; a.mode = mode;
  if (h->row_shard) {
    // the primal pass ran over this rank's rows; A' mu = sum_g A_g' mu_g and the per-rank primal norms travel in ONE all-reduce of n + 2 nranks
    // reals; the dual pass (P x, q, the summed A' mu) is then replicated and bit-identical on every rank
    real* slots = h->red_n + h->n;
    hipLaunchKernelGGL(k_rs_pack_prim, dim3(1), dim3(COSMO_BS), 0, h->stream, PARTS(h, SLOT_RP), PARTS(h, SLOT_MP), a.n_prim, slots, comm_rank(h), nr);
    CHK(launch_spmv_plain(h, h->AT, h->mu, h->red_n));
    CHK(comm_allreduce_sum(h, h->red_n, (size_t)(h->n + 2 * nr)));
    h->rs_allreduces += 1; h->rs_allreduce_elems = h->n + 2 * nr;
    prof_begin(h, KC_CHK_DUAL);
    hipLaunchKernelGGL(k_chk_dual_rs, dim3(h->P.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(h->P), h->w_prev, h->red_n, h->q, h->Dinv,
                       h->cinv, unscale ? 1 : 0, PARTS(h, SLOT_RD), PARTS(h, SLOT_MD), PARTS(h, SLOT_XPX), PARTS(h, SLOT_QX));
    prof_end(h);
    a.part_rp = slots; a.part_mp = slots + nr; a.n_prim = nr; a.n_dual = h->P.grid;
  } else {
    prof_begin(h, KC_CHK_DUAL);
    hipLaunchKernelGGL(k_chk_dual, dim3(h->PT.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(h->PT), h->w_prev,
                       h->mu, h->q, h->Dinv, h->cinv, unscale ? 1 : 0, PARTS(h, SLOT_RD), PARTS(h, SLOT_MD),
                       PARTS(h, SLOT_XPX), PARTS(h, SLOT_QX));
    prof_end(h);
  }
  a.cinv = (h->prm.unscale_residuals && h->has_scaling) ? h->cinv : 1.0;
  a.eps_abs = h->prm.eps_abs; a.eps_rel = h->prm.eps_rel;
  a.obj_true = h->prm.obj_true; a.obj_true_tol = h->prm.obj_true_tol;
  a.rho_min = h->prm.rho_min; a.rho_max = h->prm.rho_max; a.adapt_tol = h->prm.adaptive_rho_tolerance;
  a.max_adaptions = h->prm.adaptive_rho_max_adaptions;
  prof_begin(h, KC_CHK_FINAL);
  hipLaunchKernelGGL(k_chk_final, dim3(1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, a);
  prof_end(h);
  if (mode == 2) {
    prof_begin(h, KC_RHO_APPLY);
    hipLaunchKernelGGL(k_rho_apply, dim3(ew_grid(h->m > 0 ? h->m : 1)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->n,
                       h->m, h->rho_cls, h->prm.rho_min, h->prm.rho_eq_over_rho_ineq, h->mu, h->s, h->rho, h->w);
    if (h->row_shard)
      hipLaunchKernelGGL(k_rs_rho_g, dim3(ew_grid(h->m_g > 0 ? h->m_g : 1)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->m_g, h->rho_cls_g,
                         h->prm.rho_min, h->prm.rho_eq_over_rho_ineq, h->rho_g);
    CHK(refresh_op_split(h));      // rho may have changed on the device: the diagonal part of A' rho A follows (cheap, unconditional)
    prof_end(h);
  }
  h->spmv_calls[0] += 1; h->spmv_calls[1] += 1; h->spmv_calls[2] += 1;
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// timing hook: the fused operator kernel exactly as the CG iteration launches it
int32_t time_op_apply(cosmo_hip_handle* h, int reps, double* avg_seconds) {
  (void)avg_seconds;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k_op_apply, dim3((h->op_split ? h->PTm : h->PT).grid), dim3(COSMO_BS), 0, h->stream, h->ctl, 0, 2,
                       view_of(h->op_split ? h->PTm : h->PT), h->prm.sigma, h->u, h->tmp_m, h->rhs, h->r, h->c, PARTS(h, SLOT_AUX0),
                       PARTS(h, SLOT_BB), 0, 0.0, h->op_split ? h->op_diag : nullptr);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// ---- CG operator split: refresh of the rho-dependent pieces -----------------------------------------------------------
__global__ __launch_bounds__(COSMO_BS) void k_op_refresh(long long n, long long mm, const int* __restrict__ sc_ptr, const int* __restrict__ sc_row,
                                                         const real* __restrict__ sc_a2, const int* __restrict__ mrow,
                                                         const real* __restrict__ rho, real* __restrict__ diag, real* __restrict__ rho_m) {
  for (long long i = (long long)blockIdx.x * COSMO_BS + threadIdx.x; i < n + mm; i += (long long)gridDim.x * COSMO_BS) {
    if (i < n) {
      real s = 0.0;
      for (int k = sc_ptr[i]; k < sc_ptr[i + 1]; ++k) s += rho[sc_row[k]] * sc_a2[k];     // fixed order: rows ascending
      diag[i] = s;
    } else {
      rho_m[i - n] = rho[mrow[i - n]];
    }
  }
}
int32_t refresh_op_split(cosmo_hip_handle* h) {
  if (!h->op_split) return COSMO_HIP_OK;
  const long long mm = h->Am.nrows;
  hipLaunchKernelGGL(k_op_refresh, dim3(ew_grid(h->n + mm)), dim3(COSMO_BS), 0, h->stream, h->n, mm, h->op_sc_ptr, h->op_sc_row, h->op_sc_a2,
                     h->op_mrow, h->rho_g ? h->rho_g : h->rho, h->op_diag, h->op_rho_m);
  HIPCHK(h, hipGetLastError());
  return fold_refresh(h);        // the assembled operator's values follow rho_m and the diagonal
}

// launch helpers used by cg_fold.hip
int32_t launch_cg_upd(cosmo_hip_handle* h, int guard, int k, int n_uc) {
  prof_begin(h, KC_CG_UPD);
  if (h->cg_jacobi)
    hipLaunchKernelGGL(k_cg_upd<true>, dim3(ew_grid(h->n)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, h->n, PARTS(h, SLOT_UC), n_uc, h->u, h->c,
                       h->x_tl, h->r, PARTS(h, SLOT_RR), (real2*)h->cg_ru, (const real*)((FoldPlan*)h->fold)->dinv, PARTS(h, SLOT_AUX2));
  else
    hipLaunchKernelGGL(k_cg_upd<false>, dim3(ew_grid(h->n)), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, k, h->n, PARTS(h, SLOT_UC), n_uc, h->u, h->c,
                       h->x_tl, h->r, PARTS(h, SLOT_RR), (real2*)h->cg_ru, (const real*)nullptr, (real*)nullptr);
  prof_end(h);
  return COSMO_HIP_OK;
}
int32_t launch_cg_dir_check(cosmo_hip_handle* h, int guard, int kk, int n_rr) {
  prof_begin(h, KC_CG_DIR);
  hipLaunchKernelGGL(k_cg_dir, dim3(1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, kk, 1, h->n, h->n, PARTS(h, SLOT_RR), n_rr, h->r, h->u);
  prof_end(h);
  return COSMO_HIP_OK;
}

// launch helpers used by minres.hip (keeps every kernel launch next to its definition)
int32_t launch_spmv_A_rho(cosmo_hip_handle* h, int guard, int mode, const real* v, real* out) {
  // row-sharded handle: the replicated reduced operator (Am = the multi-nonzero rows of the WHOLE A with their rho; the singleton rows are the
  // diagonal the caller adds), not the rank's row slice
  const CsrDev& Ao = h->row_shard ? h->Am : h->A;
  hipLaunchKernelGGL(k_spmv_A_rho, dim3(Ao.grid > 0 ? Ao.grid : 1), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, mode, view_of(Ao), v,
                     h->row_shard ? h->op_rho_m : h->rho, out);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}
int32_t launch_reduced_rhs(cosmo_hip_handle* h, int guard, real* out_rhs) {
  if (h->row_shard) {                      // rhs = allreduce_sum(A_g' y2_g) + ls_x, as the CG path forms it (rs_enqueue_cg_rhs), then handed to the caller's buffer
    CHK(rs_enqueue_cg_rhs(h, guard));
    HIPCHK(h, hipMemcpyAsync(out_rhs, h->rhs, sizeof(real) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    return COSMO_HIP_OK;
  }
  hipLaunchKernelGGL(k_cg_rhs, dim3(h->AT.grid), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, view_of(h->AT), h->y2, h->ls_x, out_rhs,
                     PARTS(h, SLOT_BB));
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}
