import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, torch
import cosmo_jl_amd as cj
rng = np.random.default_rng(5)
for d in (160, 192):
    for nc in (50, 100, 200, 400, 800):
        sets = [cj.PsdConeTriangle(int(d * (d + 1) // 2)) for _ in range(nc)]
        m = sum(K.dim for K in sets)
        h = cj.Handle(0)
        h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
        h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
        h.project(rng.standard_normal(m))
        t, fl = h.time_psd_product(1, 20)
        ld = ((d + 63) // 64) * 64
        print("d=%d nc=%4d working set (4 matrices) %6.1f MB: %7.1f us per product, %5.1f TF/s padded, %.3f us per cone" % (d, nc, 4 * nc * ld * ld * 8 / 1e6, 1e6 * t, fl / t / 1e12, 1e6 * t / nc), flush=True)
        h.close()
