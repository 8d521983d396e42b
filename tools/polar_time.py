import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import cosmo_jl_amd as cj
rng = np.random.default_rng(5)
for d in [int(v) for v in (sys.argv[1:] or ["2000", "1000", "500", "320"])]:
    K = cj.PsdConeTriangle(d * (d + 1) // 2)
    m = K.dim
    h = cj.Handle(0)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind], [K.dim], None, None)
    G = rng.uniform(-1, 1, (d, d)); X = (G + G.T) / 2
    s = cj.problems.svec(X)
    out, rk, _ = h.project(s)
    t0 = time.perf_counter()
    for _ in range(3): out, rk, _ = h.project(s)
    t = (time.perf_counter() - t0) / 3
    w, V = np.linalg.eigh(X); Pr = (V * np.maximum(w, 0)) @ V.T
    err = np.linalg.norm(out - cj.problems.svec(Pr)) / np.linalg.norm(X)
    print("d=%d  project (incl. H2D/D2H) %.2f ms  relerr %.2e  rank %d vs %d" % (d, 1e3 * t, err, rk[0], (w > 0).sum()), flush=True)
    h.close()
