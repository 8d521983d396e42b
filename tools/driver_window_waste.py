"""Speculation waste of the Krylov budget in the DRIVER's bench window of config 5 (`bench.py --steps 20 --warmup 5`: iterations 6-25) next to the default
window (11-50): enqueued vs performed Krylov iterations, stalls, it/s."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import cosmo_jl_amd as cj
p = cj.problems.chordal_sdp()
for warm, steps in ((5, 20), (10, 40)):
    st = cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=10 ** 6, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
    cj.model.setup(md)
    h = md.handle
    h.set_iterates(md.x, md.s, md.mu); h.admm_init(); h.admm_iterate_checked(warm)
    s0 = h.get_stats(); t0 = time.perf_counter()
    h.admm_iterate_checked(steps)
    dt = time.perf_counter() - t0; s1 = h.get_stats()
    enq = s1["spmv_A"] - s0["spmv_A"]; done = s1["kkt_iters_total"] - s0["kkt_iters_total"]; solves = s1["kkt_solves"] - s0["kkt_solves"]
    print("iterations %d-%d: %.1f it/s; Krylov iterations performed %.1f per solve, enqueued %.1f per solve (no-ops %.1f = %.0f %%), stalls %d"
          % (s0["admm_iters"] + 1, s1["admm_iters"], steps / dt, done / solves, enq / solves, (enq - done) / solves, 100.0 * (enq - done) / max(enq, 1), s1["kkt_budget_stalls"] - s0["kkt_budget_stalls"]), flush=True)
    h.close()
