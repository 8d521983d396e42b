#!/usr/bin/env python
"""Times the PSD projection of the BASELINE config 5 cone mix (400 cliques, d in [20, 200]) through cosmo_hip_project."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch  # noqa
import cosmo_jl_amd as cj
rng = np.random.default_rng(5)
dk = rng.integers(20, 201, size=400)
sets = [cj.PsdConeTriangle(int(d * (d + 1) // 2)) for d in dk]
m = sum(K.dim for K in sets)
h = cj.Handle(0)
h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
s = rng.standard_normal(m)
h.set_profiling(1)
h.project(s)
t0 = time.perf_counter()
for _ in range(5):
    h.project(s)
print("mix 400 cones (m = %d): project incl. H2D/D2H copies %.2f ms; env POLAR_BATCH_MIN=%s" % (m, 1e3 * (time.perf_counter() - t0) / 5, os.environ.get("COSMO_HIP_POLAR_BATCH_MIN", "64 (default)")))
