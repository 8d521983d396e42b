"""Lab: time of the batched symmetric product of BASELINE config 5 (cosmo_hip_time_psd_product) with an alternative library build.
usage: batch_product_lab.py <lib.so | default>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
if sys.argv[1] != "default":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
p = cj.problems.chordal_sdp()
md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=2, eps_abs=0.0, eps_rel=0.0))
cj.optimize(md)
t, fl = md.handle.time_psd_product(1, 50)
print("%s: batched product %.2f us, %.1f TFLOP/s of padded tiles" % (os.path.basename(sys.argv[1]), 1e6 * t, fl / t / 1e12), flush=True)
