"""ADMM iterations/s of the accelerated loop (Anderson type-II, memory 15, safeguarded: the reference's default accelerator) against the
plain loop (EmptyAccelerator: the benchmark's setting) on the BASELINE instances, fixed work.  The accelerated loop synchronises with
the host every iteration (the accept / decline decisions of acceleration_post! are taken there), the plain loop every check_termination
iterations.  usage: accel_rate.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosmo_jl_amd as cj

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for name, gen in (("cfg5", cj.problems.chordal_sdp), ("cfg4", cj.problems.closest_correlation), ("cfg2", cj.problems.sparse_box_qp)):
    p = gen()
    for acc in (None, cj.AndersonAccelerator):
        st = cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, accelerator=acc)
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        cj.model.setup(md)
        t0 = time.perf_counter(); r = cj.optimize(md); el = time.perf_counter() - t0
        print("%s %-20s %d iterations (%d safeguarding): %.1f it/s by iter_time, status %s, r_prim %.2e" %
              (name, "Anderson(15)" if acc else "EmptyAccelerator", r.iter, getattr(r, "safeguarding_iter", 0) or 0, r.iter / r.times.iter_time, r.status, r.info.r_prim), flush=True)
