import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import cosmo_jl_amd as cj
rng = np.random.default_rng(5)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = cj.PsdConeTriangle(d * (d + 1) // 2)
h = cj.Handle(0)
h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((K.dim, 2)), np.zeros(K.dim))
h.set_cones([K.kind], [K.dim], None, None)
G = rng.uniform(-1, 1, (d, d)); X = (G + G.T) / 2
s = cj.problems.svec(X)
for _ in range(3): h.project(s)
