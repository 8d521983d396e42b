/* A plain C99 client of libcosmo_hip.so: what any non-Python / non-Julia host sees of the drop-in boundary.  On a machine without
 * a MI355X cosmo_hip_create must FAIL (COSMO_HIP_ERR_HIP) -- there is no CPU fallback -- and everything before it must work. */
#include <stdio.h>
#include <stdlib.h>
#include "cosmo_hip.h"
#include "cosmo_chordal.h"

int main(void) {
  cosmo_hip_params p;
  cosmo_hip_accel_params ap;
  cosmo_chordal_options co;
  cosmo_hip_handle* h = NULL;
  int32_t rc;
  cosmo_hip_default_params(&p);
  cosmo_hip_default_accel_params(&ap);
  cosmo_chordal_default_options(&co);
  printf("version=%d alpha=%g max_iter=%lld check_termination=%d accel_mem=%d merge=%d obj_true_is_nan=%d\n", (int)cosmo_hip_version(), p.alpha,
         (long long)p.max_iter, (int)p.check_termination, (int)ap.mem, (int)co.merge_strategy, (int)(p.obj_true != p.obj_true));
  rc = cosmo_hip_create(&h, 0);
  printf("create_rc=%d\n", (int)rc);
  if (rc == COSMO_HIP_OK) cosmo_hip_destroy(h);
  return 0;
}
