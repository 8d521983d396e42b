"""Lab: where a tile of the ragged batched product kernel spends its life (library built with -DPOLAR_LAB_TIMING, tools/build_lab_variants.sh TIMING):
shader-clock cycles of wave 0 of every workgroup, summed per phase over `reps` launches of Y = U^2 on BASELINE config 5.
usage: ragged_timing_lab.py bench/_lab/libcosmo_hip_TIMING.so"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
p = cj.problems.chordal_sdp()
md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=2, eps_abs=0.0, eps_rel=0.0))
cj.optimize(md)
lib = md.handle.lib
reps = 20
t, fl = md.handle.time_psd_product(1, reps)
out = (C.c_ulonglong * (8192 * 5))()
lib.cosmo_dbg_ragged_timing(out)
wl = (C.c_ulonglong * 8192)()
lib.cosmo_dbg_ragged_wall(wl)
a = np.frombuffer(out, dtype=np.uint64).reshape(8192, 5).astype(np.float64)
w = np.frombuffer(wl, dtype=np.uint64).astype(np.float64)
ok = a[:, 4] > 0
a, w = a[ok], w[ok]                                         # workgroups that ran a tile in the last launch
# shader clock against the 100 MHz wall clock: every tile's life on both clocks (the shader-clock counters of different XCDs are not synchronised,
# so only differences inside one workgroup mean anything)
ghz = float(np.sum(a[:, 3] - a[:, 0]) / np.sum(w)) * 0.1
print("shader clock during the launch: %.2f GHz (tile lives in s_memtime cycles over the same lives in 100 MHz s_memrealtime ticks)" % ghz)
t0 = a[:, 0].min()
pro, main, epi, whole, nk = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 3] - a[:, 0], a[:, 4]
span = a[:, 3].max() - t0
print("product %.2f us (HIP events, %d launches) = %.0f cycles at that clock; last launch: %d tiles x mean life %.0f cycles / 768 slots = %.0f cycles if perfectly packed => slot utilisation %.0f %%"
      % (1e6 * t, reps, 1e3 * t * 1e6 * ghz, len(a), whole.mean(), len(a) * whole.mean() / 768, 100 * len(a) * whole.mean() / 768 / (1e3 * t * 1e6 * ghz)))
print("  per tile, mean (median) cycles: prologue %.0f (%.0f)   main loop %.0f (%.0f) = %.0f per k-panel (%.1f panels)   epilogue %.0f (%.0f)   whole %.0f"
      % (pro.mean(), np.median(pro), main.mean(), np.median(main), main.sum() / nk.sum(), nk.mean(), epi.mean(), np.median(epi), whole.mean()))
print("  shares of a tile's life: prologue %.0f %%, main loop %.0f %%, epilogue %.0f %%" % (100 * pro.sum() / whole.sum(), 100 * main.sum() / whole.sum(), 100 * epi.sum() / whole.sum()))
for lo, hi in ((1, 3), (4, 6), (7, 9), (10, 13)):
    mk = (nk >= lo) & (nk <= hi)
    if mk.any():
        print("  tiles with %2d-%2d panels (%4d): prologue %.0f, main %.0f (%.0f per panel), epilogue %.0f" % (lo, hi, mk.sum(), pro[mk].mean(), main[mk].mean(), main[mk].sum() / nk[mk].sum(), epi[mk].mean()))
