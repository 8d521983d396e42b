#!/usr/bin/env python
"""Lab driver (round 6): the batched sign iteration's main schedule as ONE persistent dependency-driven launch (COSMO_HIP_POLAR_DATAFLOW=1,
csrc/psd_polar.hip: k_polar_dataflow) against the launch-per-product form -- iterates must be the same bits; wall time per ADMM iteration."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cosmo_jl_amd as cj

small = len(sys.argv) > 1 and sys.argv[1] == "small"
kw = dict(ncliques=40, n_total=6000, n_zero=100, n_nonneg=500) if small else {}
prob = cj.problems.chordal_sdp(**kw)
res = {}
for df in ("0", "1", "1w", "0", "1", "1w"):
    os.environ["COSMO_HIP_POLAR_DATAFLOW"] = df[0]
    os.environ["COSMO_HIP_POLAR_DATAFLOW_WGS"] = "1" if df.endswith("w") else "0"
    st = cj.Settings(max_iter=60, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, check_termination=10 ** 9)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    cj.model.setup(md)
    h = md.handle
    h.set_iterates(md.x, md.s, md.mu); h.admm_init()
    h.admm_iterate_checked(10)
    t0 = time.perf_counter(); h.admm_iterate_checked(40); dt = time.perf_counter() - t0
    w, _, s, mu = h.get_iterates()
    ps = h.polar_stats()
    print("dataflow[w = workgroup-scope counters]=%s  %.3f ms per ADMM iteration (%.1f it/s)  polar: %s" % (df, 1e3 * dt / 40, 40 / dt, {k: ps[k] for k in ("products_last_batch", "fallback_rounds", "verified", "unverified")}), flush=True)
    res.setdefault(df, []).append((w, s, mu))
    h.close()
a = res["0"][0]
for k in ("1", "1w"):
    b = res[k][0]
    print("variant %s vs launch-per-product: bit-identical iterates (w, s, mu):" % k, [bool(np.array_equal(x, y)) for x, y in zip(a, b)], " max |dw| =", float(np.max(np.abs(a[0] - b[0]))))
print("run-to-run (dataflow):", [bool(np.array_equal(x, y)) for x, y in zip(res["1"][0], res["1"][1])])
