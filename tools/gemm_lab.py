"""Lab: one large-cone PSD projection (d given) with an alternative build of the library (bench/_lab/*.so, psd_polar.hip compiled with
POLAR_LAB_* macros).  Run under `rocprofv3 --kernel-trace`; tools/gemm_lab_summary.py reads the product-kernel durations.
usage: gemm_lab.py <lib.so | default> [d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
if sys.argv[1] != "default":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(5)
K = cj.PsdConeTriangle(d * (d + 1) // 2)
h = cj.Handle(0)
h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((K.dim, 2)), np.zeros(K.dim))
h.set_cones([K.kind], [K.dim], None, None)
G = rng.uniform(-1, 1, (d, d)); s = cj.problems.svec((G + G.T) / 2)
for _ in range(4):
    out, rk, _ = h.project(s)
print(sys.argv[1], "rank", rk[0], flush=True)
