// custom.hip -- user-defined cones: the AbstractConvexSet / AbstractConvexCone plugin surface of the reference
// (src/projections.jl:4-5; required: `dim` and project!(x, C); optional: in_dual, in_pol_recc --
// docs/src/literate/custom_cone.jl:9-17, 62-68).  project!(::SplitVector, ::CompositeConvexSet) (src/convexset.jl:885-891)
// hands every set its contiguous slice of s; for a user cone that slice is staged device -> pinned host, the user's callback
// projects it in place on the calling thread, and it is copied back, in stream order with the kernels that project the other
// cones.  A composite set with a custom cone therefore costs one stream synchronisation per iteration -- the price of a host
// plugin inside a device-resident loop; every built-in cone stays on the device.
#include "internal.h"
#include <vector>

void custom_free(cosmo_hip_handle* h) {
  h->custom.clear();
  if (h->custom_host) { (void)hipHostFree(h->custom_host); h->custom_host = nullptr; }
  if (h->custom_halt) { (void)hipHostFree(h->custom_halt); h->custom_halt = nullptr; }
}

int32_t custom_plan_create(cosmo_hip_handle* h) {
  custom_free(h);
  const ConeTable& C = h->cones;
  long long tot = 0;
  for (size_t k = 0; k < C.type.size(); ++k) {
    if (C.type[k] != COSMO_HIP_CUSTOM) continue;
    CustomCone cc;
    cc.cone = (long long)k; cc.off = C.off[k]; cc.dim = C.dim[k]; cc.host_off = tot;
    tot += cc.dim;
    h->custom.push_back(cc);
  }
  if (h->custom.empty()) return COSMO_HIP_OK;
  HIPCHK(h, hipHostMalloc((void**)&h->custom_host, sizeof(real) * (size_t)(tot > 0 ? tot : 1), hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc((void**)&h->custom_halt, sizeof(int), hipHostMallocDefault));
  *h->custom_halt = 0;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_set_custom_cone(cosmo_hip_handle* h, int64_t cone, cosmo_hip_project_fn project, cosmo_hip_cone_test_fn in_dual,
                                             cosmo_hip_cone_test_fn in_pol_recc, void* user) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (!h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_custom_cone: set_cones must be called first");
  if (!project) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_custom_cone: a projection callback is required");
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "set_custom_cone: not available on a row-sharded handle (user-defined cones are projected on the host)");
  for (CustomCone& cc : h->custom) {
    if (cc.cone != cone) continue;
    cc.project = project; cc.in_dual = in_dual; cc.in_pol_recc = in_pol_recc; cc.user = user;
    return COSMO_HIP_OK;
  }
  return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_custom_cone: cone %lld is not of type COSMO_HIP_CUSTOM", (long long)cone);
}

// guard != 0: inside the loop; once a status has been decided on the device (ctl->halt) the iterates are frozen, so the
// callback is skipped as every loop kernel is
int32_t custom_enqueue_project(cosmo_hip_handle* h, real* s, int guard) {
  if (h->custom.empty()) return COSMO_HIP_OK;
  for (const CustomCone& cc : h->custom) {
    if (!cc.project) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "custom cone %lld has no projection callback (cosmo_hip_set_custom_cone)", cc.cone);
    if (cc.dim > 0)
      HIPCHK(h, hipMemcpyAsync(h->custom_host + cc.host_off, s + cc.off, sizeof(real) * (size_t)cc.dim, hipMemcpyDeviceToHost, h->stream));
  }
  if (guard) HIPCHK(h, hipMemcpyAsync(h->custom_halt, &h->ctl->halt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (guard && *h->custom_halt) return COSMO_HIP_OK;
  for (const CustomCone& cc : h->custom) cc.project(h->custom_host + cc.host_off, (int64_t)cc.dim, cc.user);
  for (const CustomCone& cc : h->custom)
    if (cc.dim > 0)
      HIPCHK(h, hipMemcpyAsync(s + cc.off, h->custom_host + cc.host_off, sizeof(real) * (size_t)cc.dim, hipMemcpyHostToDevice, h->stream));
  return COSMO_HIP_OK;
}

// Membership tests of the infeasibility certificates (src/infeasibility.jl:21-23, 60-62) for the custom cones.
//   which 0: support_function!(dyn, cone, tol) = in_dual(-dyn) ? 0 : Inf  (src/convexset.jl:933-936)  -> the callback sees -v
//   which 1: in_pol_recc(v, cone, tol)                                                                -> the callback sees v
// A cone without the callback never certifies.
int32_t custom_test(cosmo_hip_handle* h, const real* v_dev, int which, real tol, bool* ok) {
  if (h->custom.empty() || !*ok) return COSMO_HIP_OK;
  std::vector<real> buf;
  for (const CustomCone& cc : h->custom) {
    const cosmo_hip_cone_test_fn fn = which == 0 ? cc.in_dual : cc.in_pol_recc;
    if (!fn) { *ok = false; return COSMO_HIP_OK; }
    buf.resize((size_t)(cc.dim > 0 ? cc.dim : 1));
    if (cc.dim > 0) {
      HIPCHK(h, hipMemcpyAsync(buf.data(), v_dev + cc.off, sizeof(real) * (size_t)cc.dim, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (which == 0) for (long long i = 0; i < cc.dim; ++i) buf[(size_t)i] = -buf[(size_t)i];
    if (!fn(buf.data(), (int64_t)cc.dim, (double)tol, cc.user)) { *ok = false; return COSMO_HIP_OK; }
  }
  return COSMO_HIP_OK;
}
