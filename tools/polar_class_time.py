#!/usr/bin/env python
"""Throughput of the batched symmetric-product kernel of the sign iteration (csrc/psd_polar.hip: k_symm_gemm_batch) on UNIFORM batches:
N cones of one side d each.  Separates what the cone mix of BASELINE config 5 costs because of padding (d rounded up to 64-wide tiles)
from what short inner products cost (a d = 64 cone has 4 k-panels per tile).  Uses the library's own timing hook
(cosmo_hip_time_psd_product: HIP events around back-to-back launches, flop = PADDED tile flops).

usage: polar_class_time.py [ncones=400]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch  # noqa
import cosmo_jl_amd as cj

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(5)
for d in (32, 64, 65, 96, 128, 129, 160, 192, 193, 200, 256):
    sets = [cj.PsdConeTriangle(int(d * (d + 1) // 2)) for _ in range(nc)]
    m = sum(K.dim for K in sets)
    h = cj.Handle(0)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
    h.project(rng.standard_normal(m))
    t, fl = h.time_psd_product(1, 20)
    useful = nc * float(d) ** 3
    print("d = %3d x %d cones: %7.1f us per product, %5.1f TFLOP/s of padded tiles, %5.1f TFLOP/s of d^3 (padding factor %.2f)"
          % (d, nc, 1e6 * t, fl / t / 1e12, useful / t / 1e12, fl / useful), flush=True)
    h.close()
