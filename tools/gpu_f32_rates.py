#!/usr/bin/env python
"""ADMM iterations/s of BASELINE configs 2 / 4 / 5 in Float32 (libcosmo_hip_f32.so) next to Float64, fixed work (eps = 0, N iterations).
The Krylov tolerance schedule 1 / k^1.5 drops below what Float32 can resolve after a few dozen solves; the CG then runs to its
stagnation point every time, so the Float32 Krylov counts are NOT comparable with Float64 ones -- iterations/s are reported with the
mean Krylov count next to them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, gen in (("cfg4", cj.problems.closest_correlation), ("cfg5", cj.problems.chordal_sdp), ("cfg2", cj.problems.sparse_box_qp)):
    p = gen()
    for dt in (np.float64, np.float32):
        st = cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
        md = cj.Model(dtype=dt); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        r = cj.optimize(md)
        print("%s %s: %.1f it/s (%d iterations, %.1f Krylov its per solve, obj %.6e, polar %s)" % (name, np.dtype(dt).name, r.iter / r.times.iter_time, r.iter,
              r.kkt_iters_total / (r.iter + 1.0), r.obj_val, {k: md.handle.polar_stats()[k] for k in ("fallback_rounds", "unverified")}), flush=True)
