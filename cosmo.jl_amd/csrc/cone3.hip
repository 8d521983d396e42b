// cone3.hip -- the 3-dimensional non-symmetric cones: ExponentialCone, PowerCone and their duals
// (src/convexset.jl:497-779).  These are scalar algorithms per 3-vector (bisection + Newton for K_exp, Newton on Hien's
// equation for K_pow), so the device form is one thread per cone; each thread follows the reference's branch order and
// iteration limits exactly -- only exp/log/pow come from the device math library instead of openlibm, which moves results
// by a few ulp (the algorithms themselves stop at 1e-8).  Dual cones use the Moreau decomposition as the reference does.
#include "device_utils.h"
#include <math.h>
#include <vector>

namespace {

struct V3 { double x, y, z; };

// ---- K_exp ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool exp_in_cone(V3 v, double tol) {        // convexset.jl:589-594
  return (v.y > 0.0 && v.y * exp(v.x / v.y) <= v.z + tol) || (v.x <= tol && v.y == 0.0 && v.z >= -tol);
}
__device__ __forceinline__ bool exp_in_dual(V3 v, double tol) {        // :596-601
  return (v.x < 0.0 && -v.x * exp(v.y / v.x) - 2.718281828459045 * v.z <= tol) || (fabs(v.x) <= tol && v.y >= -tol && v.z >= -tol);
}
__device__ double exp_find_min_t(double lam, double s0, double t0, double tol) {   // :570-587
  double dt = fmax(-t0, tol);
  for (int k = 0; k < 150; ++k) {
    const double f = dt * (dt + t0) / (lam * lam) - s0 / lam + log(dt / lam) + 1.0;
    const double grad_f = (2.0 * dt + t0) / (lam * lam) + 1.0 / dt;
    dt = dt - f / grad_f;
    if (dt <= -t0) { dt = -t0; break; }
    else if (dt <= 0.0) { dt = 0.0; break; }
    else if (fabs(f) < tol) break;
  }
  return dt + t0;
}
__device__ __forceinline__ double exp_grad_dual(double lam, V3& v, V3 v0, double tol) {   // :555-568
  v.z = exp_find_min_t(lam, v0.y, v0.z, tol);
  v.y = (1.0 / lam) * (v.z - v0.z) * v.z;
  v.x = v0.x - lam;
  return (v.y == 0.0) ? v.x : v.x + v.y * log(v.y / v.z);
}
// returns the case 1..4 of project!(::ExponentialCone) (:510-537)
__device__ int exp_project(V3& v, int max_iter, double tol) {
  if (exp_in_cone(v, 0.0)) return 1;
  if (exp_in_dual(V3{-v.x, -v.y, -v.z}, 0.0)) { v = V3{0.0, 0.0, 0.0}; return 2; }
  if (v.x < 0.0 && v.y < 0.0) { v.y = 0.0; v.z = fmax(v.z, 0.0); return 3; }
  const V3 v0 = v;
  double l = 0.0, lam = 0.125;                                          // get_bisection_bounds (:542-553)
  double g = exp_grad_dual(lam, v, v0, tol);
  int guard = 0;
  while (g > 0.0 && guard++ < 2000) { l = lam; lam *= 2.0; g = exp_grad_dual(lam, v, v0, tol); }
  double u = lam;
  for (int k = 0; k < max_iter; ++k) {                                   // project_exp! (:540-553)
    lam = (u + l) / 2.0;
    g = exp_grad_dual(lam, v, v0, tol);
    if (g > 0.0) l = lam; else u = lam;
    if (u - l < tol) break;
  }
  return 4;
}

// ---- K_pow(alpha) ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pow_in_cone(V3 v, double a, double tol) {   // :707-713
  return v.x >= 0.0 && v.y >= 0.0 && pow(v.x, a) * pow(v.y, 1.0 - a) >= fabs(v.z) - tol;
}
__device__ __forceinline__ bool pow_in_dual(V3 v, double a, double tol) {   // :716-722
  return v.x >= -tol && v.y >= -tol && pow(v.x, a) * pow(v.y, 1.0 - a) >= fabs(v.z) * pow(a, a) * pow(1.0 - a, 1.0 - a) - tol;
}
__device__ __forceinline__ double pow_phic(double c0, double az, double r, double a) {   // :686-688
  return fmax(0.5 * (c0 + sqrt(c0 * c0 + 4.0 * a * r * (az - r))), 1e-10);
}
__device__ int pow_project(V3& v, double a, int max_iter, double tol) {      // :626-684
  if (pow_in_cone(v, a, 0.0)) return 1;
  if (pow_in_dual(V3{-v.x, -v.y, -v.z}, a, 0.0)) { v = V3{0.0, 0.0, 0.0}; return 2; }
  if (fabs(v.z) <= tol) { v.x = fmax(v.x, 0.0); v.y = fmax(v.y, 0.0); return 3; }
  const double x0 = v.x, y0 = v.y, z0 = v.z, az = fabs(v.z);
  double r = az / 2.0, phix = 0.0, phiy = 0.0;
  for (int k = 0; k < max_iter; ++k) {
    phix = pow_phic(x0, az, r, a);
    phiy = pow_phic(y0, az, r, 1.0 - a);
    const double prod = pow(phix, a) * pow(phiy, 1.0 - a);
    const double phi = prod - r;
    if (fabs(phi) < tol) break;
    const double dphix = a / (2.0 * phix - x0) * (az - 2.0 * r);
    const double dphiy = (1.0 - a) / (2.0 * phiy - y0) * (az - 2.0 * r);
    const double dphi = prod * (a * dphix / phix + (1.0 - a) * dphiy / phiy) - 1.0;
    r = r - phi / dphi;
    r = fmin(fmax(r, 0.0), az);
  }
  v.x = phix; v.y = phiy; v.z = z0 * r / az;
  return 4;
}

__device__ __forceinline__ int project_kind(V3& v, int kind, double a, int it_exp, int it_pow, double tol_exp, double tol_pow) {
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
__device__ __forceinline__ bool in_dual_kind(V3 x, int kind, double a, double tol) {
  switch (kind) {
    case COSMO_HIP_EXP: return exp_in_dual(x, tol);
    case COSMO_HIP_DUAL_EXP: return exp_in_cone(x, tol);
    case COSMO_HIP_POW: return pow_in_dual(x, a, tol);
    default: return pow_in_cone(x, a, tol);
  }
}

__global__ __launch_bounds__(COSMO_BS) void k_cone3_project(const Ctl* __restrict__ ctl, int guard, int nc, const int* __restrict__ off,
                                                            const int* __restrict__ kind, const double* __restrict__ alpha,
                                                            double* __restrict__ s, int* __restrict__ branch) {
  if (guard && ctl->halt) return;
  for (int c = blockIdx.x * COSMO_BS + threadIdx.x; c < nc; c += gridDim.x * COSMO_BS) {
    double* p = s + off[c];
    V3 v{p[0], p[1], p[2]};
    const int br = project_kind(v, kind[c], alpha[c], 100, 20, 1e-8, 1e-8);
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
    branch[c] = br;
  }
}

// infeasibility certificates: both support_function!(dyn) (in_dual(-dyn)) and in_pol_recc(adx) (= in_dual(-adx)) test
// the NEGATED vector against the dual cone (convexset.jl:603-605, 724-726, 772, 928-936)
__global__ __launch_bounds__(COSMO_BS) void k_cone3_in_dual_neg(int nc, const int* __restrict__ off, const int* __restrict__ kind,
                                                                const double* __restrict__ alpha, const double* __restrict__ v,
                                                                double tol, int* __restrict__ flag) {
  int viol = 0;
  for (int c = blockIdx.x * COSMO_BS + threadIdx.x; c < nc; c += gridDim.x * COSMO_BS) {
    const double* p = v + off[c];
    if (!in_dual_kind(V3{-p[0], -p[1], -p[2]}, kind[c], alpha[c], tol)) viol = 1;
  }
  if (viol) atomicOr(flag, 1);
}

template <class T>
int32_t dev_upload(cosmo_hip_handle* h, T** dst, const std::vector<T>& src) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  if (src.empty()) return COSMO_HIP_OK;
  HIPCHK(h, hipMalloc((void**)dst, sizeof(T) * src.size()));
  HIPCHK(h, hipMemcpyAsync(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}

}  // namespace

void cone3_free(cosmo_hip_handle* h) {
  if (h->c3_off) (void)hipFree(h->c3_off);
  if (h->c3_kind) (void)hipFree(h->c3_kind);
  if (h->c3_alpha) (void)hipFree(h->c3_alpha);
  if (h->c3_branch) (void)hipFree(h->c3_branch);
  h->c3_off = h->c3_kind = h->c3_branch = nullptr; h->c3_alpha = nullptr; h->ncone3 = 0;
  h->c3_cone_index.clear();
}

// table of the 3-d cones; they are projected redundantly on every rank of a sharded run (O(1) work per cone)
int32_t cone3_plan_create(cosmo_hip_handle* h) {
  cone3_free(h);
  const ConeTable& C = h->cones;
  std::vector<int> off, kind; std::vector<double> alpha;
  for (size_t k = 0; k < C.type.size(); ++k) {
    if (C.type[k] < COSMO_HIP_EXP || C.type[k] > COSMO_HIP_DUAL_POW) continue;
    off.push_back((int)C.off[k]); kind.push_back(C.type[k]); alpha.push_back(C.param.empty() ? 0.0 : C.param[k]);
    h->c3_cone_index.push_back((int)k);
  }
  h->ncone3 = (int)off.size();
  if (!h->ncone3) return COSMO_HIP_OK;
  CHK(dev_upload(h, &h->c3_off, off)); CHK(dev_upload(h, &h->c3_kind, kind)); CHK(dev_upload(h, &h->c3_alpha, alpha));
  std::vector<int> zeros(off.size(), 0);
  CHK(dev_upload(h, &h->c3_branch, zeros));
  return COSMO_HIP_OK;
}

int32_t cone3_enqueue_project(cosmo_hip_handle* h, double* s, int guard) {
  if (!h->ncone3) return COSMO_HIP_OK;
  int g = (h->ncone3 + COSMO_BS - 1) / COSMO_BS;
  if (g > 4096) g = 4096;
  prof_begin(h, KC_SOC);
  hipLaunchKernelGGL(k_cone3_project, dim3(g), dim3(COSMO_BS), 0, h->stream, h->ctl, guard, h->ncone3, h->c3_off, h->c3_kind, h->c3_alpha, s,
                     h->c3_branch);
  prof_end(h);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

// sets *flag (device int) when some 3-d cone fails in_dual(-v)
int32_t cone3_enqueue_in_dual_neg(cosmo_hip_handle* h, const double* v, double tol, int* flag) {
  if (!h->ncone3) return COSMO_HIP_OK;
  int g = (h->ncone3 + COSMO_BS - 1) / COSMO_BS;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_cone3_in_dual_neg, dim3(g), dim3(COSMO_BS), 0, h->stream, h->ncone3, h->c3_off, h->c3_kind, h->c3_alpha, v, tol, flag);
  HIPCHK(h, hipGetLastError());
  return COSMO_HIP_OK;
}

int32_t cone3_get_branches(cosmo_hip_handle* h, int32_t* out_per_cone) {
  if (!h->ncone3) return COSMO_HIP_OK;
  std::vector<int> br((size_t)h->ncone3);
  HIPCHK(h, hipMemcpyAsync(br.data(), h->c3_branch, sizeof(int) * br.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->ncone3; ++i) out_per_cone[h->c3_cone_index[i]] = br[i];
  return COSMO_HIP_OK;
}
