#!/usr/bin/env python
"""The PRODUCTION SpMV kernels (k_spmv_plain on A, k_cg_rhs-shaped A' product, the fused [P | A'] operator kernel k_op_apply) on
matrices of BASELINE config 2's size (n = 100k, m = 200k, nnz(A) = 2M, nnz(P) = 500k) whose column patterns have locality:
  random   : uniformly random columns (the cfg2 generator)             -- every gathered double touches its own cache line
  banded   : row i holds 10 nonzeros within a band of width 256 around column i n / m
  blocked  : rows in groups of 512 draw their columns from one cluster of 1024 consecutive columns
Answers VERDICT r1 item 6: is the kernel or the gather pattern the limiter?  Prints one line per (pattern, kernel) with the
algorithmic bytes, the average launch time (HIP events around 200 back-to-back launches) and the fraction of 8 TB/s."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import cosmo_jl_amd as cj

n, m, per_row = 100_000, 200_000, 10


def pattern(kind, rng):
    rows = np.repeat(np.arange(m), per_row)
    if kind == "random":
        cols = rng.integers(0, n, size=m * per_row)
    elif kind == "banded":
        centre = (np.arange(m) * n // m)[:, None]
        cols = np.clip(centre + rng.integers(-128, 128, size=(m, per_row)), 0, n - 1).ravel()
    else:
        base = rng.integers(0, n - 1024, size=(m + 511) // 512)
        cols = (base[np.arange(m) // 512][:, None] + rng.integers(0, 1024, size=(m, per_row))).ravel()
    A = sp.coo_matrix((rng.standard_normal(m * per_row), (rows, cols)), shape=(m, n)).tocsc()
    A.sort_indices()
    return A


def pmatrix(kind, rng):
    k = 2 * n
    i = rng.integers(0, n, size=k)
    j = rng.integers(0, n, size=k) if kind == "random" else np.clip(i + rng.integers(-64, 64, size=k), 0, n - 1)
    S = sp.coo_matrix((0.1 * rng.standard_normal(k), (i, j)), shape=(n, n)).tocsc()
    S = (S + S.T).tocsc()
    P = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0)).tocsc()
    P.sort_indices()
    return P


for kind in ("random", "banded", "blocked"):
    rng = np.random.default_rng(7)
    A, P = pattern(kind, rng), pmatrix(kind, rng)
    h = cj.Handle(0)
    h.set_problem(P, np.zeros(n), A, np.zeros(m))
    for name, which in (("A x (k_spmv_plain)", cj._ffi.MAT_A), ("A' y (k_spmv_plain)", cj._ffi.MAT_AT), ("P x (k_spmv_plain)", cj._ffi.MAT_P),
                        ("[P | A'] fused operator (k_op_apply)", cj._ffi.MAT_OP)):
        t, b = h.time_spmv(which, 200)
        print(json.dumps(dict(pattern=kind, kernel=name, nnzA=int(A.nnz), nnzP=int(P.nnz), algorithmic_MB=round(b / 1e6, 2), avg_launch_us=round(t * 1e6, 2),
                              GBps=round(b / t / 1e9, 1), frac_of_8TBps=round(b / t / 8e12, 3))), flush=True)
    h.close()
