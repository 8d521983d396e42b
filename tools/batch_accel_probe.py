#!/usr/bin/env python
"""Accelerated loop on the reference's dual-infeasible LP family: single-problem device path vs batch path (status, iterations, accelerator counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
from tests import infeasible_instances as INF

TIGHT = dict(tol_constant=1e-10, tol_exponent=0.0)
kinds = {INF.ZERO: cj.ZeroSet, INF.NONNEG: cj.Nonnegatives, INF.SOC: cj.SecondOrderCone}
for fam in ("dual_infeasible_1", "primal_infeasible_1"):
    for seed in (1, 2, 3):
        gen, accepted, _ = INF.FAMILIES[fam]
        P, q, cons = gen(seed)
        st = cj.Settings(accelerator=cj.AndersonAccelerator, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT), max_iter=2000, eps_abs=1e-5, eps_rel=1e-5)
        md = cj.Model(); cj.assemble(md, P, q, [cj.Constraint(A, b, kinds[k]) for (A, b, k, d) in cons], settings=st)
        r1 = cj.optimize(md)
        a1 = md.handle.accel_stats()
        md2 = cj.Model(); cj.assemble(md2, P, q, [cj.Constraint(A, b, kinds[k]) for (A, b, k, d) in cons], settings=st)
        B, _ = cj.model.prepare_batch([md2], 0)
        r2 = B.optimize()[0]
        a2 = {k: int(v[0]) for k, v in B.accel_stats().items()}
        B.close()
        print(fam, seed, "single:", r1.status, r1.iter, a1, "| batch:", cj._ffi.STATUS_NAMES[r2.status], r2.iter, a2, flush=True)
