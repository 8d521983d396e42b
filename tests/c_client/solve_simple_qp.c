/* A plain C99 client that SOLVES a problem through include/cosmo_hip.h on the GPU: the reference's simple QP
 * (/root/reference/test/UnitTests/simple.jl:22-31: P = [4 1; 1 2], q = [1; 1], l <= A x <= u with A = [1 1; 1 0; 0 1], l = [1; 0; 0],
 * u = [1; 0.7; 0.7], written as two Nonnegatives constraints; expected :45-47: status :Solved, x = [0.3; 0.7], obj_val = 1.88, atol 1e-3).
 * Everything a C compiler sees of the boundary is exercised: the struct layouts of cosmo_hip_params / cosmo_hip_result, Julia's
 * SparseMatrixCSC{Float64, Int64} argument order (1-based colptr / rowval), the call sequence
 *     create -> set_problem -> set_cones -> scale_ruiz -> set_params -> set_iterates -> optimize -> get_iterates.
 * Built for both libraries: -DCOSMO_HIP_REAL_FLOAT + -lcosmo_hip_f32 is the COSMO.Model{Float32} instantiation.
 * Output (one line, parsed by tests/test_gpu_c_client.py):  status=1 iter=.. x0=.. x1=.. obj=.. kkt="..."                                  */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "cosmo_hip.h"

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    int32_t rc_ = (call);                                                                                    \
    if (rc_ != COSMO_HIP_OK) {                                                                               \
      fprintf(stderr, "%s failed: rc=%d (%s)\n", #call, (int)rc_, h ? cosmo_hip_last_error(h) : "no handle"); \
      return 10 + (int)rc_;                                                                                  \
    }                                                                                                        \
  } while (0)

int main(void) {
  cosmo_hip_handle* h = NULL;
  /* A x + s = b, s in K (COSMO's internal form, src/constraint.jl: Constraint(A_c, b_c, K) means A_c x + b_c in K, stored as A = -A_c, b = b_c):
   *   Constraint(-A, u, Nonnegatives):  rows 1-3 =  A,  b = u        (s = u - A x >= 0)
   *   Constraint( A, -l, Nonnegatives): rows 4-6 = -A,  b = -l       (s = A x - l >= 0)                                                     */
  const int64_t n = 2, m = 6;
  /* P = [4 1; 1 2], column-major CSC, 1-based */
  const int64_t P_colptr[3] = {1, 3, 5};
  const int64_t P_rowval[4] = {1, 2, 1, 2};
  const cosmo_hip_real P_nzval[4] = {4, 1, 1, 2};
  /* A (6 x 2): column 1 has rows {1, 2, 4, 5}, column 2 rows {1, 3, 4, 6} */
  const int64_t A_colptr[3] = {1, 5, 9};
  const int64_t A_rowval[8] = {1, 2, 4, 5, 1, 3, 4, 6};
  const cosmo_hip_real A_nzval[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  const cosmo_hip_real q[2] = {1, 1};
  const cosmo_hip_real b[6] = {1, (cosmo_hip_real)0.7, (cosmo_hip_real)0.7, -1, 0, 0};
  const int32_t cone_type[2] = {COSMO_HIP_NONNEG, COSMO_HIP_NONNEG};
  const int64_t cone_dim[2] = {3, 3};
  cosmo_hip_real D[2], E[6], w[8], w_prev[8], s[6], mu[6];
  double c = 1.0;
  cosmo_hip_params prm;
  cosmo_hip_result res;
  double x0, x1, obj;

  CHECK(cosmo_hip_create(&h, 0));
  CHECK(cosmo_hip_set_problem(h, n, m, P_colptr, P_rowval, P_nzval, A_colptr, A_rowval, A_nzval, q, b));
  CHECK(cosmo_hip_set_cones(h, 2, cone_type, cone_dim, NULL, NULL));
  CHECK(cosmo_hip_scale_ruiz(h, 10, 1e-4, 1e4, D, E, &c));               /* settings.scaling = 10, MIN_SCALING, MAX_SCALING (src/settings.jl, src/scaling.jl:21-116) */
  cosmo_hip_default_params(&prm);                                          /* COSMO.Settings() defaults; kkt_kind CG = CGIndirectKKTSolver */
#ifdef COSMO_HIP_REAL_FLOAT
  prm.eps_abs = 1e-4; prm.eps_rel = 1e-4;                                  /* Float32: the reference's unit tests loosen the tolerances the same way */
#endif
  CHECK(cosmo_hip_set_params(h, &prm, NULL));
  CHECK(cosmo_hip_set_iterates(h, NULL, NULL, NULL));
  CHECK(cosmo_hip_optimize(h, &res));
  CHECK(cosmo_hip_get_iterates(h, w, w_prev, s, mu));
  /* reverse_scaling! (src/scaling.jl:170-179): x = D .* x_scaled; the objective of the unscaled problem from the unscaled x */
  x0 = (double)D[0] * (double)w[0]; x1 = (double)D[1] * (double)w[1];
  obj = 0.5 * (4 * x0 * x0 + 2 * x0 * x1 + 2 * x1 * x1) + x0 + x1;
  printf("status=%d iter=%lld x0=%.6f x1=%.6f obj=%.6f cost=%.6f rho_updates=%d kkt=\"%s\"\n", (int)res.status, (long long)res.iter, x0, x1, obj, res.cost,
         (int)res.n_rho_updates, cosmo_hip_kkt_recurrence(h));
  CHECK(cosmo_hip_destroy(h));
  if (res.status != COSMO_HIP_SOLVED) return 2;
  if (fabs(x0 - 0.3) > 1e-3 || fabs(x1 - 0.7) > 1e-3 || fabs(obj - 1.88) > 1e-3 || fabs(res.cost - 1.88) > 1e-3) return 3;
  return 0;
}
