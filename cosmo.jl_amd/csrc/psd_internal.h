// psd_internal.h -- declarations shared by psd.hip (Jacobi eigensolvers) and psd_polar.hip (polar-iteration projection).
#pragma once
#include "device_utils.h"
#include <vector>

typedef real v4d __attribute__((ext_vector_type(4)));
// The 16x16x4 matrix instruction of the build's scalar type: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] in both, but the
// four results of a lane sit in DIFFERENT rows: v_mfma_f64_16x16x4_f64 -> rows (l >> 4) + 4 r, v_mfma_f32_16x16x4_f32 -> rows
// 4 (l >> 4) + r (column l & 15 in both).  ACC_ROW(lane, r) is that map; every piece of code that owns a 16x16 tile "in the C
// layout" (the wave-level Jacobi, the SYRK / symmetric-product epilogues, the panel update) indexes rows through it.
#if REAL_IS_FLOAT
#define MFMA_REAL(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define ACC_ROW(lane, r) (4 * ((lane) >> 4) + (r))
__device__ __forceinline__ real hw_sqrt(real x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ real hw_rcp(real x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ real hw_rsq(real x) { return __builtin_amdgcn_rsqf(x); }
#else
#define MFMA_REAL(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define ACC_ROW(lane, r) (((lane) >> 4) + 4 * (r))
__device__ __forceinline__ real hw_sqrt(real x) { return __builtin_amdgcn_sqrt(x); }
__device__ __forceinline__ real hw_rcp(real x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ real hw_rsq(real x) { return __builtin_amdgcn_rsq(x); }
#endif
#define WLD 17  // leading dimension of the 16x16 LDS tiles (padded against bank conflicts)
#define PSD_MAX_SWEEPS 40
#define PSD_EPS REAL_EPS

struct PsdConeDev {
  int off;        // first row of the cone in s
  int d;          // matrix side
  int kind;       // COSMO_HIP_PSD_SQUARE / COSMO_HIP_PSD_TRIANGLE
  int ld;         // leading dimension of G (multiple of 16 >= d)
  int ncp;        // padded number of columns: nb * 8, nb even
  int nb;         // number of 8-column blocks (even)
  long long goff; // offset of G in the workspace (doubles)
  int coff;       // offset of the per-column arrays
  int cone_index; // index in the composite set
};

struct PsdPlan {
  std::vector<PsdConeDev> cones;       // all PSD cones with d > 1
  std::vector<int> tiny, wg, large;    // indices into `cones` by size class
  std::vector<int> wg_waves;           // waves per workgroup for every wg-class launch group
  std::vector<std::vector<int>> wg_groups;
  PsdConeDev* d_cones = nullptr;
  int *d_tiny = nullptr, *d_large = nullptr;
  std::vector<int*> d_wg_groups;
  // projection-time launch groups of the workgroup Jacobi kernels: the wg-class cones that are NOT handled by the batched
  // matrix-sign path (psd_polar.hip); the full wg_groups stay in use for the definiteness tests of the certificates
  std::vector<int> polar_batch;        // indices of the wg-class cones projected by the batched matrix-sign path
  bool large_by_polar = false;         // the real large cones are projected by psd_polar.hip
  std::vector<int> cplx;               // complex Hermitian cones (real 2r x 2r embedding, matrix-sign paths only)
  std::vector<int> pj_waves;
  std::vector<std::vector<int>> pj_groups;
  std::vector<int*> d_pj_groups;
  real* G = nullptr;
  real* colw = nullptr;              // per column: sigma then scale factor
  real* cshift = nullptr;            // per cone shift c
  int* rank = nullptr;                 // per cone nnz_lambda
  int* flags = nullptr;                // [0] sweep-rotated flag (large path), [1] error flag
  real* eigmin = nullptr;            // per cone smallest eigenvalue (definiteness tests)
  long long gsize = 0;
  int ncolw = 0;
  int last_large_sweeps = 0;
  real tol_factor = 0.125;   // rotate while |w_pq| > tol_factor * d * eps * sqrt(w_pp w_qq)
  int dbg = 0;
};

