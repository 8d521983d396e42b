// cone3.hip -- the 3-dimensional non-symmetric cones: ExponentialCone, PowerCone and their duals
// (src/convexset.jl:497-779).  These are scalar algorithms per 3-vector (bisection + Newton for K_exp, Newton on Hien's
// equation for K_pow), so the device form is one thread per cone; each thread follows the reference's branch order and
// iteration limits exactly -- only exp/log/pow come from the device math library instead of openlibm, which moves results
// by a few ulp (the algorithms themselves stop at 1e-8).  Dual cones use the Moreau decomposition as the reference does.
#include "device_utils.h"
#include "cone3.h"
#include <math.h>
#include <vector>

using namespace cone3;

namespace {

__global__ __launch_bounds__(COSMO_BS) void k_cone3_project(const Ctl* __restrict__ ctl, int guard, int nc, const int* __restrict__ off,
                                                            const int* __restrict__ kind, const real* __restrict__ alpha,
                                                            real* __restrict__ s, int* __restrict__ branch) {
  if (guard && ctl->halt) return;
  for (int c = blockIdx.x * COSMO_BS + threadIdx.x; c < nc; c += gridDim.x * COSMO_BS) {
    real* p = s + off[c];
    V3 v{p[0], p[1], p[2]};
    const int br = project_kind(v, kind[c], alpha[c], 100, 20, 1e-8, 1e-8);
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
    branch[c] = br;
  }
}

// infeasibility certificates: both support_function!(dyn) (in_dual(-dyn)) and in_pol_recc(adx) (= in_dual(-adx)) test
// the NEGATED vector against the dual cone (convexset.jl:603-605, 724-726, 772, 928-936)
__global__ __launch_bounds__(COSMO_BS) void k_cone3_in_dual_neg(int nc, const int* __restrict__ off, const int* __restrict__ kind,
                                                                const real* __restrict__ alpha, const real* __restrict__ v,
                                                                real tol, int* __restrict__ flag) {
  int viol = 0;
  for (int c = blockIdx.x * COSMO_BS + threadIdx.x; c < nc; c += gridDim.x * COSMO_BS) {
    const real* p = v + off[c];
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
  std::vector<int> off, kind; std::vector<real> alpha;
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

int32_t cone3_enqueue_project(cosmo_hip_handle* h, real* s, int guard) {
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
int32_t cone3_enqueue_in_dual_neg(cosmo_hip_handle* h, const real* v, real tol, int* flag) {
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
