#!/usr/bin/env python
"""Lab (round 6): what a Krylov iteration of BASELINE config 5 would cost if the 6 000 ten-nonzero Zero / Nonnegatives rows of A (600 k of the 708 k entries of the
assembled operator M = P + sigma I + A' rho A) were applied in factored form instead of being assembled.  Probe: the same instance WITHOUT those rows (M has only P, the
diagonal and the two-nonzero consensus rows), timed by cosmo_hip_time_krylov as the loop enqueues it (captured chain)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cosmo_jl_amd as cj

for label, kw in (("cfg5", {}), ("cfg5 without the Zero / Nonneg rows", dict(n_zero=1, n_nonneg=1))):
    prob = cj.problems.chordal_sdp(**kw)
    st = cj.Settings(max_iter=10 ** 6, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, check_termination=10 ** 9)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    cj.model.setup(md)
    h = md.handle
    h.set_iterates(md.x, md.s, md.mu); h.admm_init(); h.admm_iterate_checked(3)
    t, b, nl = min(h.time_krylov(200) for _ in range(3))
    print("%-40s nnz(M) = %8d  %6.2f us per Krylov iteration (%d launches), %s" % (label, h.fold_stats()["nnz"], 1e6 * t, nl, h.kkt_recurrence()), flush=True)
    h.close()
