#!/usr/bin/env python
"""Run one BASELINE configuration end to end on the GPU for a fixed number of ADMM iterations and (optionally) time the
(the CPU comparison lives in bench.py's cpu_baseline).  Prints one JSON line.   python tools/run_config.py cfg5 --iters 50"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import cosmo_jl_amd as cj

ap = argparse.ArgumentParser()
ap.add_argument("config", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--d", type=int, default=2000)
args = ap.parse_args()
t0 = time.time()
if args.config == "cfg3":
    probs = [cj.problems.socp(seed=1000 + k) for k in range(1024)]
    st = cj.Settings(max_iter=args.iters, eps_abs=0.0, eps_rel=0.0)
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    tg = time.time() - t0
    res = cj.optimize_batch(mods)
    out = dict(config="cfg3", problems=len(res), iters=args.iters, gen_s=round(tg, 2), iter_time_s=res[0].times.iter_time,
               batch_iters_per_s=args.iters / res[0].times.iter_time, problem_iters_per_s=args.iters * len(res) / res[0].times.iter_time,
               mean_cg_per_iter=float(np.mean([r.kkt_iters_total / (r.iter + 1) for r in res])))
    prob = probs[0]
else:
    prob = {"cfg1": cj.problems.dense_qp, "cfg2": cj.problems.sparse_box_qp, "cfg4": lambda: cj.problems.closest_correlation(d=args.d),
            "cfg5": cj.problems.chordal_sdp}[args.config]()
    tg = time.time() - t0
    st = cj.Settings(max_iter=args.iters, eps_abs=0.0, eps_rel=0.0)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    t1 = time.time(); cj.model.setup(md); ts = time.time() - t1
    h = md.handle
    h.set_iterates(md.x, md.s, md.mu)
    r = h.optimize()
    out = dict(config=args.config, n=md.n, m=md.m, nnzA=int(md.A.nnz), cones=len(md.sets), iters=int(r.iter), gen_s=round(tg, 2), setup_s=round(ts, 2),
               iter_time_s=r.iter_time, iters_per_s=r.iter / r.iter_time, mean_cg_per_iter=r.kkt_iters_total / max(1, r.kkt_solves),
               r_prim=r.r_prim, r_dual=r.r_dual, psd=h.psd_stats())
# (the CPU side of every configuration is bench.py's `cpu_baseline` leg: tools/ never touches oracle/)
print(json.dumps(out))
