// minres.hip -- MINRES KKT solvers (placeholder)
#include "internal.h"
int32_t minres_alloc(cosmo_hip_handle* h) { return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "MINRES not built yet"); }
int32_t minres_enqueue_solve(cosmo_hip_handle* h, int guard, bool from_loop) { (void)guard; (void)from_loop; return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "MINRES not built yet"); }
