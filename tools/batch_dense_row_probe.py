#!/usr/bin/env python
"""Lab: what ONE dense row of A (a budget constraint sum(x) = 1, as in the reference's portfolio examples) costs the persistent batch kernels, whose
sparse passes give every row to one thread: 256 problems of BASELINE config 3 with and without such a row, per-iteration cost of each kernel form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import cosmo_jl_amd as cj

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 256
large = len(sys.argv) > 2 and sys.argv[2] == "large"        # n = 800, m = 1600: the <512, 2, 4> register kernel (index-order assignment)
if large:
    base = [cj.problems.socp(n=800, m=1600, ncones=40, nnz=8000, seed=1000 + k) for k in range(nprob)]
else:
    base = [cj.problems.socp(n=300, m=600, ncones=30, nnz=6000, seed=1000 + k) for k in range(nprob)]      # config-3 structure at 60 % size: the image has room for the extra row


def with_dense_row(p):
    n = p["A"].shape[1]
    A = sp.vstack([p["A"], sp.csr_matrix(np.ones((1, n)))]).tocsc()
    return dict(P=p["P"], q=p["q"], A=A, b=np.concatenate([p["b"], [1.0]]), sets=list(p["sets"]) + [cj.ZeroSet(1)])


def run(probs, label, env):
    for k, v in env.items():
        os.environ[k] = v
    far = dict(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 6)
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(**far)); mods.append(md)
    B, _ = cj.model.prepare_batch(mods, 0)
    ki = B.kernel_info()
    B.iterate(10, with_init=True)
    _, _, k0 = B.counters()
    t0 = time.perf_counter(); B.iterate(100); dt = time.perf_counter() - t0
    _, _, k1 = B.counters()
    print("%-44s %8.1f us per batch iteration, Krylov per problem-iteration mean %.1f / max %.1f   [%s%s]"
          % (label, 1e6 * dt / 100, (k1 - k0).mean() / 100, (k1 - k0).max() / 100, ki["form"], ", sliced" if ki["sliced"] else ""), flush=True)
    B.close()
    for k in env:
        os.environ.pop(k, None)


dense = [with_dense_row(p) for p in base]
name = "socp 800 x 1600" if large else "socp 300 x 600"
for probs, tag in ((base, name), (dense, name + " + one dense row")):
    run(probs, tag + ", default kernel", {})
    if probs is dense:
        run(probs, tag + ", one thread per row (COSMO_HIP_BATCH_LONG=0)", {"COSMO_HIP_BATCH_LONG": "0"})
    run(probs, tag + ", LDS-image (generic loops)", {"COSMO_HIP_BATCH_REG": "0"})
    run(probs, tag + ", streaming", {"COSMO_HIP_BATCH_LDS": "0"})
