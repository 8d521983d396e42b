"""Generates tests/golden/baseline_convergent.json: BASELINE.json configs 4 (closest correlation, d = 2000) and 5 (decomposed chordal
SDP, 400 cliques, n = 50 000) at FULL size, solved by the CPU oracle with the reference's DEFAULT settings (eps_abs = eps_rel = 1e-5,
adaptive rho, check_termination = 25, Ruiz scaling; CG indirect KKT solver with the 1 / k^1.5 tolerance schedule) until `Solved`.
Pins what a user sees end to end -- status, iteration count, objective, the sequence of rho updates -- where baseline_cfg4/5.npz pin a
few tight-CG iterations entry by entry.  Like those, this pins the RESTATED algorithm (the Julia reference cannot run here).
cfg4 takes ~5 minutes, cfg5 ~2-3 hours of CPU (25 s per ADMM iteration in NumPy).  Usage: python tests/golden/make_fixtures_convergent.py [cfg4|cfg5]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cosmo_jl_amd as cj            # noqa: E402  (problem generators only; no device code is touched)
from oracle import cosmo_oracle as O  # noqa: E402
from tests import util               # noqa: E402

# cfg5 does not reach eps = 1e-5 within a practical CPU budget (the device run is still at r_prim = 2.6e-2 after 700 iterations): its
# fixture is the state after 150 iterations of the default schedule (status Max_iter_reached, residuals, objective, rho updates)
MAX_ITER = {"cfg2": 1000, "cfg4": 400, "cfg5": 150}
OUT = os.path.join(ROOT, "tests", "golden", "baseline_convergent.json")


def problem(name):
    if name == "cfg2":
        return cj.problems.sparse_box_qp()
    return cj.problems.closest_correlation() if name == "cfg4" else cj.problems.chordal_sdp()


def solve_cfg5_to_convergence(p):
    """Round 3: cfg5 DOES converge with the default settings -- the device needs 2825 iterations (9.5 s, tools/cfg5_convergent_device.py) -- and
    the compiled restatement now projects PSD cones (LAPACK dsyevr + dsyrk), so the CPU side of that solve is affordable: ~1.5-2 hours on one core
    (161 k Krylov iterations on a 2.9 M-nonzero A).  Stored under the key "cfg5_solved"; the 150-iteration entry "cfg5" stays."""
    from oracle import cosmo_oracle_c as OC
    ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg", max_iter=6000))
    return OC.run(ws, native=os.path.exists(os.path.join(ROOT, "oracle", "_build", "libcosmo_oracle_c_native.so")))


def solve_cfg2(p):
    """cfg2 (n = 100 000, Box cone, ~550 Krylov iterations per solve) is out of reach of the NumPy loop; the COMPILED restatement of the
    same loop (oracle/cosmo_oracle_c.c, pinned to the NumPy oracle by tests/test_oracle_c.py) runs it: setup (Ruiz scaling, rho vector) by
    the NumPy oracle, the loop in C."""
    from oracle import cosmo_oracle_c as OC
    ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg", max_iter=MAX_ITER["cfg2"]))
    r = OC.run(ws, native=os.path.exists(os.path.join(ROOT, "oracle", "_build", "libcosmo_oracle_c_native.so")))
    return r


if __name__ == "__main__":
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in (sys.argv[1:] or ["cfg4", "cfg5"]):
        t0 = time.time()
        p = problem(name) if name != "cfg5_solved" else None
        if name == "cfg5_solved":
            p = problem("cfg5")
            r = solve_cfg5_to_convergence(p)
            out[name] = dict(max_iter=6000, status=r["status"], iter=int(r["iter"]), obj_val=float(r["obj_val"]), r_prim=float(r["r_prim"]), r_dual=float(r["r_dual"]),
                             rho_updates=[float(v) for v in r["rho_updates"]], cg_iters_total=int(r["cg_iters_total"]), x_norm=float(np.linalg.norm(r["x"])),
                             x_absmax=float(np.max(np.abs(r["x"]))), oracle_seconds=round(time.time() - t0, 1), proj_seconds=round(r["proj_time"], 1),
                             oracle="compiled C restatement (oracle/cosmo_oracle_c.c; dsyevr + dsyrk projections)")
            print(name, out[name], flush=True)
            json.dump(out, open(OUT, "w"), indent=1)
            continue
        if name == "cfg2":
            r = solve_cfg2(p)
            out[name] = dict(max_iter=MAX_ITER[name], status=r["status"], iter=int(r["iter"]), obj_val=float(r["obj_val"]), r_prim=float(r["r_prim"]),
                             r_dual=float(r["r_dual"]), rho_updates=[float(v) for v in r["rho_updates"]], cg_iters_total=int(r["cg_iters_total"]),
                             x_norm=float(np.linalg.norm(r["x"])), x_absmax=float(np.max(np.abs(r["x"]))), oracle_seconds=round(time.time() - t0, 1),
                             oracle="compiled C restatement (oracle/cosmo_oracle_c.c)")
            print(name, out[name], flush=True)
            json.dump(out, open(OUT, "w"), indent=1)
            continue
        ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg", max_iter=MAX_ITER[name]))
        r = ws.optimize()
        out[name] = dict(max_iter=MAX_ITER[name], status=r.status, iter=int(r.iter), obj_val=float(r.obj_val), r_prim=float(r.r_prim), r_dual=float(r.r_dual),
                         rho_updates=[float(v) for v in r.rho_updates], cg_iters_total=int(np.sum(r.cg_iters)),
                         x_norm=float(np.linalg.norm(r.x)), x_absmax=float(np.max(np.abs(r.x))), oracle_seconds=round(time.time() - t0, 1))
        print(name, out[name], flush=True)
        json.dump(out, open(OUT, "w"), indent=1)
