import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cosmo_jl_amd as cj
if os.environ.get("COSMO_LAB_LIB"):
    cj._ffi.LIB_PATH = os.environ["COSMO_LAB_LIB"]
probs = [cj.problems.socp(seed=1000 + k) for k in range(256)]
far = dict(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 6)
def run(label, env, st=None):
    for k, v in env.items(): os.environ[k] = v
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st or cj.Settings(**far)); mods.append(md)
    B, _ = cj.model.prepare_batch(mods, 0)
    ki = B.kernel_info()
    B.iterate(10, with_init=True)
    t0 = time.perf_counter(); B.iterate(200); dt = time.perf_counter() - t0
    print("%-28s %.1f us per batch iteration   %s" % (label, 1e6 * dt / 200, ki), flush=True)
    B.close()
    for k in env: os.environ.pop(k, None)
which = sys.argv[1:] or ["reg", "rcg", "aa"]
if "reg" in which: run("reg", {})
if "rcg" in which: run("lds rcg", {"COSMO_HIP_BATCH_REG": "0", "COSMO_HIP_BATCH_EXT": "1"})
if "aa" in which: run("lds rcg + Anderson", {"COSMO_HIP_BATCH_REG": "0"}, cj.Settings(accelerator=cj.AndersonAccelerator, **far))
