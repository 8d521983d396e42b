#!/usr/bin/env python
"""Per-iteration cost of the accelerated batch loop: 256 problems of BASELINE config 3 (one per CU), 200 iterations through batch_iterate (no certificates,
no exits), for the register kernel, the LDS-image kernel and the LDS-image kernel with the accelerator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
probs = [cj.problems.socp(seed=1000 + k) for k in range(nprob)]


def run(st, label, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    B, _ = cj.model.prepare_batch(mods, 0)
    ki = B.kernel_info()
    B.iterate(10, with_init=True)
    _, _, k0 = B.counters()
    t0 = time.perf_counter(); B.iterate(iters); dt = time.perf_counter() - t0
    _, _, k1 = B.counters()
    a = B.accel_stats()
    kk = (k1 - k0)
    print("%-34s %d iterations: %.1f ms = %.1f us per batch iteration; Krylov iterations per problem-iteration mean %.1f / max %.1f; accelerated %d declined %d"
          % (label, iters, 1e3 * dt, 1e6 * dt / iters, kk.mean() / iters, kk.max() / iters, a["accelerated"].sum(), a["declined"].sum()), flush=True)
    print("%-34s   kernel: %s%s, %d registers and %d scratch bytes per thread in the loaded code object, %d + %d LDS bytes%s"
          % ("", ki["form"], " (sliced image)" if ki["sliced"] else "", ki["registers"], ki["scratch_bytes"], ki["lds_bytes"], ki["static_lds_bytes"],
             ", length-sorted compute assignment" if ki["sorted_assignment"] else ""), flush=True)
    RATES[label] = 1e6 * dt / iters
    B.close()
    for k in (env or {}):
        os.environ.pop(k, None)


RATES = {}
far = dict(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 6)
AA = cj.Settings(accelerator=cj.AndersonAccelerator, **far)
# "LDS-image" rows: the instantiation that serves batches with PSD / exponential / power cones (COSMO_HIP_BATCH_EXT=1 selects it on this cone-free batch;
# the accelerated LDS-image kernel IS that instantiation), with the generic Krylov loop of round 4 (COSMO_HIP_BATCH_LDSCG=0) and the register-CG form
run(cj.Settings(**far), "register kernel")
run(cj.Settings(**far), "LDS-image, generic CG", {"COSMO_HIP_BATCH_REG": "0", "COSMO_HIP_BATCH_EXT": "1", "COSMO_HIP_BATCH_LDSCG": "0"})
run(cj.Settings(**far), "LDS-image, register CG", {"COSMO_HIP_BATCH_REG": "0", "COSMO_HIP_BATCH_EXT": "1"})
run(AA, "register + Anderson")
run(cj.Settings(accelerator=cj.with_options(cj.AndersonAccelerator, mem=5), **far), "register + Anderson(5)")
run(AA, "LDS-image + Anderson, generic CG", {"COSMO_HIP_BATCH_REG": "0", "COSMO_HIP_BATCH_LDSCG": "0"})
run(AA, "LDS-image + Anderson, register CG", {"COSMO_HIP_BATCH_REG": "0"})
print("ratios: LDS-image (register CG) / register kernel = %.2f; accelerated: LDS-image / register kernel = %.2f"
      % (RATES["LDS-image, register CG"] / RATES["register kernel"], RATES["LDS-image + Anderson, register CG"] / RATES["register + Anderson"]))
