"""Lab: per-phase shader-clock cycles of workgroup 0 in the register-resident batch kernel (library built with -DCOSMO_BATCH_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
probs = [cj.problems.socp(seed=seed0 + k) for k in range(8)]
st = cj.Settings(max_iter=200, eps_abs=0.0, eps_rel=0.0)
mods = []
for p in probs:
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
res = cj.optimize_batch(mods)
lib = cj.load_library()
out = (ctypes.c_longlong * 8)()
lib.cosmo_dbg_batch_timing(out)
v = list(out)
ncg = max(v[3], 1)
print("problem 0: iters", v[5], "CG its", v[3], "kkt total (host)", res[0].kkt_iters_total, "time", res[0].times.iter_time)
print("cycles per CG it: A pass %.0f  PT pass %.0f  bsum %.0f ; kernel total cycles %d (%.0f per CG it)" % (v[0] / ncg, v[1] / ncg, v[2] / ncg, v[4], v[4] / ncg))
