// psd.hip -- PSD cone projections (placeholder until the Jacobi eigensolver lands in this file)
#include "internal.h"
struct PsdPlan { int ncones = 0; };
int32_t psd_plan_create(cosmo_hip_handle* h) {
  int n = 0;
  for (size_t k = 0; k < h->cones.type.size(); ++k)
    if ((h->cones.type[k] == COSMO_HIP_PSD_SQUARE || h->cones.type[k] == COSMO_HIP_PSD_TRIANGLE) && h->cones.dim[k] > 1) ++n;
  if (n) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "PSD cones not built yet");
  return COSMO_HIP_OK;
}
void psd_plan_destroy(cosmo_hip_handle* h) { (void)h; }
int32_t psd_enqueue_project(cosmo_hip_handle* h, double* s, bool guard) { (void)h; (void)s; (void)guard; return COSMO_HIP_OK; }
int32_t psd_get_ranks(cosmo_hip_handle* h, int64_t* r) { (void)h; (void)r; return COSMO_HIP_OK; }
