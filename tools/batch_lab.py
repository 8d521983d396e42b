"""Lab: BASELINE config 3 (or a prefix of it) through a given build of libcosmo_hip.so -- batch-iterations/s of the timed window, the slowest problem's
microseconds per Krylov iteration, and a hash of the iterates (lab variants of the register kernel must reproduce the production bits).
usage: batch_lab.py <lib.so | default> [nprob=1024] [steps=100] [warmup=25]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
from cosmo_jl_amd import _ffi
if sys.argv[1] != "default":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 25
probs = [cj.problems.socp(seed=1000 + k) for k in range(nprob)]
st = cj.Settings(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 9, check_infeasibility=10 ** 9)
mods = []
for p in probs:
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
B, _ = cj.model.prepare_batch(mods, 0)
B.iterate(warm, with_init=True)
_, _, k0 = B.counters()
best = None
t0 = time.perf_counter(); B.iterate(steps); el = time.perf_counter() - t0
_, _, k1 = B.counters()
kry = (k1 - k0)
h = hashlib.sha256()
for k in range(0, nprob, max(1, nprob // 64)):
    h.update(B.get_iterates(k)[0].tobytes())
print("%s: %d problems, %.1f batch-it/s, Krylov per problem mean %.1f max %d, %.3f us per Krylov iteration of the slowest, iterate hash %s"
      % (os.path.basename(sys.argv[1]), nprob, steps / el, kry.mean(), kry.max(), 1e6 * el / max(kry.max(), 1), h.hexdigest()[:16]), flush=True)
B.close()
