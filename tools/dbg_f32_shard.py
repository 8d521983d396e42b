import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
import cosmo_jl_amd as cj
for dtype in (np.float32, np.float64):
    prob = cj.problems.chordal_sdp(ncliques=8, dmin=4, dmax=30, sep_min=1, sep_max=3, n_total=600, n_zero=5, n_nonneg=10)
    st = cj.Settings(max_iter=100, eps_abs=0, eps_rel=0)
    out = []
    for mode in ("ref", "comm", "ref2"):
        model = cj.Model(dtype=dtype); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
        if mode == "comm":
            cj.model.setup(model)
            model.handle.comm_init(0, 1, cj.Handle.comm_unique_id())
            model.handle.set_cone_shard(cj.partition_cones_contiguous(cj.cone_costs(model.sets), 1))
        res = cj.optimize(model)
        ps = model.handle.polar_stats()
        out.append(res.x.copy())
        print(dtype.__name__, mode, "iter", res.iter, "kkt", res.kkt_iters_total, {k: ps[k] for k in ("batch_cones", "schedule_steps", "fallback_rounds", "verified", "unverified", "projections", "err_max_e18")})
    print("  ref==comm", np.array_equal(out[0], out[1]), "ref==ref2", np.array_equal(out[0], out[2]), "maxdiff", np.max(np.abs(out[0] - out[1])))
