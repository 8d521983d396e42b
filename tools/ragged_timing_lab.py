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
a = np.frombuffer(out, dtype=np.uint64).reshape(8192, 5).astype(np.float64)
a = a[a[:, 4] > 0]                                          # workgroups that ran a tile in the last launch
t0 = a[:, 0].min()
pro, main, epi, whole, nk = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 3] - a[:, 0], a[:, 4]
span = a[:, 3].max() - t0
print("product %.2f us (HIP events, %d launches); last launch: %d tiles, first start -> last end %.0f shader-clock cycles" % (1e6 * t, reps, len(a), span))
print("  per tile, mean (median) cycles: prologue %.0f (%.0f)   main loop %.0f (%.0f) = %.0f per k-panel (%.1f panels)   epilogue %.0f (%.0f)   whole %.0f"
      % (pro.mean(), np.median(pro), main.mean(), np.median(main), main.sum() / nk.sum(), nk.mean(), epi.mean(), np.median(epi), whole.mean()))
print("  shares of a tile's life: prologue %.0f %%, main loop %.0f %%, epilogue %.0f %%" % (100 * pro.sum() / whole.sum(), 100 * main.sum() / whole.sum(), 100 * epi.sum() / whole.sum()))
st = a[:, 0] - t0
print("  tile start times: %.0f %% of the tiles start in the first 5 %% of the span, the last tile starts at %.0f %% of it" % (100 * np.mean(st < 0.05 * span), 100 * st.max() / span))
for lo, hi in ((1, 3), (4, 6), (7, 9), (10, 13)):
    mk = (nk >= lo) & (nk <= hi)
    if mk.any():
        print("  tiles with %2d-%2d panels (%4d): prologue %.0f, main %.0f (%.0f per panel), epilogue %.0f" % (lo, hi, mk.sum(), pro[mk].mean(), main[mk].mean(), main[mk].sum() / nk[mk].sum(), epi[mk].mean()))
