#!/usr/bin/env python
"""Wall time of fixed-iteration ADMM runs with the single-launch persistent CG (csrc/cg_persist.hip) on and off, small operators."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cosmo_jl_amd as cj

PROBS = {
    "box_qp_n3000_nnz50k": lambda: cj.problems.sparse_box_qp(n=3000, m=6000, nnz=50000, seed=9),
    "box_qp_n1000_nnz15k": lambda: cj.problems.sparse_box_qp(n=1000, m=2000, nnz=15000, seed=9),
    "box_qp_n6000_nnz100k": lambda: cj.problems.sparse_box_qp(n=6000, m=12000, nnz=100000, seed=9),
    "socp_n500": lambda: cj.problems.socp(seed=1000),
    "chordal_small": lambda: cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=70, sep_min=1, sep_max=3, n_total=2500, n_zero=40, n_nonneg=80),
}
out = {}
for name, gen in PROBS.items():
    p = gen()
    row = {}
    for persist in ("1", "0"):
        os.environ["COSMO_HIP_CG_PERSIST"] = persist
        st = cj.Settings(max_iter=300, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        r = cj.optimize(md)
        md2 = cj.Model(); md2.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        r = cj.optimize(md2)                                   # second run: warm caches / JIT-free
        s = md2.handle.cg_persist_stats()
        row["persist" + persist] = dict(iter_time_ms=round(1e3 * r.times.iter_time, 2), kkt_iters=r.kkt_iters_total, enabled=s["enabled"], launches=s["launches"],
                                        us_per_krylov_iter=round(1e6 * r.times.iter_time / max(1, r.kkt_iters_total), 2))
    out[name] = row
    print(name, json.dumps(row), flush=True)
