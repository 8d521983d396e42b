"""How many of the speculatively enqueued Krylov iterations of a BASELINE configuration (default cfg5; argument cfg2 | cfg4) are no-ops behind
a converged solve: enqueued iterations (host counter of the product launches) against performed ones (device counter), windows of 40 iterations
after 10 (cfg5's bench window is iterations 11-50)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import cosmo_jl_amd as cj
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
p = {"cfg5": cj.problems.chordal_sdp, "cfg2": cj.problems.sparse_box_qp, "cfg4": cj.problems.closest_correlation}[which]()
st = cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=10 ** 6, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, check_termination=10 ** 9)
md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
cj.model.setup(md)
h = md.handle
h.set_iterates(md.x, md.s, md.mu); h.admm_init(); h.admm_iterate_checked(10)
for window in (40, 40, 40):
    s0 = h.get_stats(); t0 = time.perf_counter()
    h.admm_iterate_checked(window)
    dt = time.perf_counter() - t0; s1 = h.get_stats()
    enq = s1["spmv_A"] - s0["spmv_A"]; done = s1["kkt_iters_total"] - s0["kkt_iters_total"]; solves = s1["kkt_solves"] - s0["kkt_solves"]
    print("iterations %d-%d: %.1f it/s; Krylov iterations performed %.1f per solve, enqueued %.1f per solve (no-ops %.1f = %.0f %%), stalls %d"
          % (s0["admm_iters"] + 1, s1["admm_iters"], window / dt, done / solves, enq / solves, (enq - done) / solves, 100.0 * (enq - done) / max(enq, 1), s1["kkt_budget_stalls"] - s0["kkt_budget_stalls"]))
