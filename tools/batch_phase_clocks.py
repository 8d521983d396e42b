"""Lab (round 4): in-kernel shader-clock cycles of the register batch kernel's phases per Krylov iteration (library built with -DCOSMO_BATCH_TIMING;
workgroup 0 is instrumented), for a typical problem of BASELINE config 3 and for its slowest problems (largest Krylov counts in iterations 26-125).
usage: COSMO_LAB_LIB=bench/_lab/libcosmo_hip_TIMING.so python tools/batch_phase_clocks.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
if os.environ.get("COSMO_LAB_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["COSMO_LAB_LIB"])
st = cj.Settings(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 9, check_infeasibility=10 ** 9)


def batch(seeds):
    mods = []
    for sd in seeds:
        p = cj.problems.socp(seed=sd)
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    return cj.model.prepare_batch(mods, 0)[0]


def clocks():
    out = (ctypes.c_longlong * 8)()
    cj.load_library().cosmo_dbg_batch_timing(out)
    return np.array(list(out), dtype=np.int64)


B = batch(range(1000, 2024))
B.iterate(25, with_init=True)
_, _, k0 = B.counters()
B.iterate(100)
_, _, k1 = B.counters()
kry = k1 - k0
top = np.argsort(kry)[::-1][:4]
print("largest Krylov counts in iterations 26-125:", [(int(i), int(kry[i])) for i in top], "median", int(np.median(kry)))
B.close()
for label, idx in [("typical (problem 0)", 0)] + [("straggler (problem %d)" % i, int(i)) for i in top[:2]]:
    Bs = batch([1000 + idx] + [1000 + ((idx + 1 + j) % 1024) for j in range(7)])       # workgroup 0 = the problem of interest
    c0 = clocks()
    Bs.iterate(25, with_init=True); Bs.iterate(100)
    c = clocks() - c0
    n = max(int(c[3]), 1)
    print("%-28s Krylov iterations %6d: head (beta, u, publish, barrier) %5.0f, A pass %6.0f, column pass (P + A') %6.0f, u'c reduction %5.0f, tail (alpha, x / r update, r'r sum, sqrt) %5.0f "
          "cycles per Krylov iteration; whole kernel %6.0f cycles per Krylov iteration (%d ADMM iterations)" % (label, n, c[6] / n, c[0] / n, c[1] / n, c[2] / n, c[7] / n, c[4] / n, int(c[5])), flush=True)
    Bs.close()
