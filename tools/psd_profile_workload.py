#!/usr/bin/env python
"""PSD projection workloads for rocprofv3 runs: `mix` = 400 cliques d in [20,200] (the cone mix of BASELINE config 5),
`d200` = 256 cones of side 200, `d2000` = one cone of side 2000 (config 4).  Two projections each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch  # noqa
import cosmo_jl_amd as cj
rng = np.random.default_rng(5)
which = sys.argv[1]
if which == "mix":
    dk = rng.integers(20, 201, size=400)
    sets = [cj.PsdConeTriangle(int(d * (d + 1) // 2)) for d in dk]
elif which == "d200":
    sets = [cj.PsdConeTriangle(200 * 201 // 2)] * 256
else:
    sets = [cj.PsdConeTriangle(2000 * 2001 // 2)]
m = sum(K.dim for K in sets)
h = cj.Handle(0)
h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
s = rng.standard_normal(m)
h.project(s); h.project(s)
print(which, h.psd_stats())
