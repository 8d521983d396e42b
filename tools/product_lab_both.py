"""Lab: time of the sign iteration's product kernel with an alternative library build -- the batched product of BASELINE config 5 and the d = 2000 product of
config 4 (cosmo_hip_time_psd_product: HIP events around back-to-back launches).  usage: product_lab_both.py <lib.so | default>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
if sys.argv[1] != "default":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
for name, p, which in (("config 5 batched product", cj.problems.chordal_sdp(), 1), ("config 4 d = 2000 product", cj.problems.closest_correlation(), 0)):
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=2, eps_abs=0.0, eps_rel=0.0))
    cj.optimize(md)
    md.handle.time_psd_product(which, 200)
    ts = [md.handle.time_psd_product(which, 100)[0] for _ in range(3)]
    print("%-24s %-28s %.2f us (runs: %s)" % (os.path.basename(sys.argv[1]), name, 1e6 * min(ts), ", ".join("%.2f" % (1e6 * t) for t in ts)), flush=True)
