// cone3.h -- device routines of the 3-dimensional non-symmetric cones (ExponentialCone, PowerCone and their duals, src/convexset.jl:497-779):
// one thread per cone, the reference's branch order and iteration limits.  Shared by csrc/cone3.hip (single-problem path: one launch over all such
// cones) and csrc/batch.hip (batch mode: the problem's persistent workgroup projects its own cones).
#pragma once
#include "device_utils.h"
#include <math.h>

namespace cone3 {

struct V3 { real x, y, z; };

// ---- K_exp ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool exp_in_cone(V3 v, real tol) {        // convexset.jl:589-594
  return (v.y > R(0.0) && v.y * exp(v.x / v.y) <= v.z + tol) || (v.x <= tol && v.y == R(0.0) && v.z >= -tol);
}
__device__ __forceinline__ bool exp_in_dual(V3 v, real tol) {        // :596-601
  return (v.x < R(0.0) && -v.x * exp(v.y / v.x) - R(2.718281828459045) * v.z <= tol) || (fabs(v.x) <= tol && v.y >= -tol && v.z >= -tol);
}
static __device__ real exp_find_min_t(real lam, real s0, real t0, real tol) {   // :570-587
  real dt = fmax(-t0, tol);
  for (int k = 0; k < 150; ++k) {
    const real f = dt * (dt + t0) / (lam * lam) - s0 / lam + log(dt / lam) + R(1.0);
    const real grad_f = (R(2.0) * dt + t0) / (lam * lam) + R(1.0) / dt;
    dt = dt - f / grad_f;
    if (dt <= -t0) { dt = -t0; break; }
    else if (dt <= R(0.0)) { dt = R(0.0); break; }
    else if (fabs(f) < tol) break;
  }
  return dt + t0;
}
__device__ __forceinline__ real exp_grad_dual(real lam, V3& v, V3 v0, real tol) {   // :555-568
  v.z = exp_find_min_t(lam, v0.y, v0.z, tol);
  v.y = (R(1.0) / lam) * (v.z - v0.z) * v.z;
  v.x = v0.x - lam;
  return (v.y == R(0.0)) ? v.x : v.x + v.y * log(v.y / v.z);
}
// returns the case 1..4 of project!(::ExponentialCone) (:510-537)
static __device__ int exp_project(V3& v, int max_iter, real tol) {
  if (exp_in_cone(v, 0.0)) return 1;
  if (exp_in_dual(V3{-v.x, -v.y, -v.z}, 0.0)) { v = V3{0.0, 0.0, 0.0}; return 2; }
  if (v.x < R(0.0) && v.y < R(0.0)) { v.y = R(0.0); v.z = fmax(v.z, R(0.0)); return 3; }
  const V3 v0 = v;
  real l = 0.0, lam = 0.125;                                          // get_bisection_bounds (:542-553)
  real g = exp_grad_dual(lam, v, v0, tol);
  int guard = 0;
  while (g > R(0.0) && guard++ < 2000) { l = lam; lam *= R(2.0); g = exp_grad_dual(lam, v, v0, tol); }
  real u = lam;
  for (int k = 0; k < max_iter; ++k) {                                   // project_exp! (:540-553)
    lam = (u + l) / R(2.0);
    g = exp_grad_dual(lam, v, v0, tol);
    if (g > R(0.0)) l = lam; else u = lam;
    if (u - l < tol) break;
  }
  return 4;
}

// ---- K_pow(alpha) ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pow_in_cone(V3 v, real a, real tol) {   // :707-713
  return v.x >= R(0.0) && v.y >= R(0.0) && pow(v.x, a) * pow(v.y, R(1.0) - a) >= fabs(v.z) - tol;
}
__device__ __forceinline__ bool pow_in_dual(V3 v, real a, real tol) {   // :716-722
  return v.x >= -tol && v.y >= -tol && pow(v.x, a) * pow(v.y, R(1.0) - a) >= fabs(v.z) * pow(a, a) * pow(R(1.0) - a, R(1.0) - a) - tol;
}
__device__ __forceinline__ real pow_phic(real c0, real az, real r, real a) {   // :686-688
  return fmax(R(0.5) * (c0 + sqrt(c0 * c0 + R(4.0) * a * r * (az - r))), R(1e-10));
}
static __device__ int pow_project(V3& v, real a, int max_iter, real tol) {      // :626-684
  if (pow_in_cone(v, a, 0.0)) return 1;
  if (pow_in_dual(V3{-v.x, -v.y, -v.z}, a, 0.0)) { v = V3{0.0, 0.0, 0.0}; return 2; }
  if (fabs(v.z) <= tol) { v.x = fmax(v.x, 0.0); v.y = fmax(v.y, 0.0); return 3; }
  const real x0 = v.x, y0 = v.y, z0 = v.z, az = fabs(v.z);
  real r = az / R(2.0), phix = R(0.0), phiy = R(0.0);
  for (int k = 0; k < max_iter; ++k) {
    phix = pow_phic(x0, az, r, a);
    phiy = pow_phic(y0, az, r, R(1.0) - a);
    const real prod = pow(phix, a) * pow(phiy, R(1.0) - a);
    const real phi = prod - r;
    if (fabs(phi) < tol) break;
    const real dphix = a / (R(2.0) * phix - x0) * (az - R(2.0) * r);
    const real dphiy = (R(1.0) - a) / (R(2.0) * phiy - y0) * (az - R(2.0) * r);
    const real dphi = prod * (a * dphix / phix + (R(1.0) - a) * dphiy / phiy) - R(1.0);
    r = r - phi / dphi;
    r = fmin(fmax(r, 0.0), az);
  }
  v.x = phix; v.y = phiy; v.z = z0 * r / az;
  return 4;
}

__device__ __forceinline__ int project_kind(V3& v, int kind, real a, int it_exp, int it_pow, real tol_exp, real tol_pow) {
  switch (kind) {
    case COSMO_HIP_EXP: return exp_project(v, it_exp, tol_exp);
    case COSMO_HIP_POW: return pow_project(v, a, it_pow, tol_pow);
    default: {                                                            // dual cones: v + Proj_K(-v)   (:774-779)
      const V3 v0 = v;
      V3 t{-v.x, -v.y, -v.z};
      const int c = (kind == COSMO_HIP_DUAL_EXP) ? exp_project(t, it_exp, tol_exp) : pow_project(t, a, it_pow, tol_pow);
      v = V3{t.x + v0.x, t.y + v0.y, t.z + v0.z};
      return c;
    }
  }
}

// in_dual(x, cone, tol) of the composite-set element (dual cones: dual of the dual = primal, :770-772)
__device__ __forceinline__ bool in_dual_kind(V3 x, int kind, real a, real tol) {
  switch (kind) {
    case COSMO_HIP_EXP: return exp_in_dual(x, tol);
    case COSMO_HIP_DUAL_EXP: return exp_in_cone(x, tol);
    case COSMO_HIP_POW: return pow_in_dual(x, a, tol);
    default: return pow_in_cone(x, a, tol);
  }
}


}  // namespace cone3
