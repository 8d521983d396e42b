#!/usr/bin/env python
"""bench.py -- ADMM iterations/sec (fp64) of the MI355X-native COSMO hot path on BASELINE.json's metric config.

Workload (BASELINE.json configs[1]): random sparse QP n=100k, m=200k, nnz(A)=2M, one Box cone,
CG indirect KKT, fp64.  A "step" is ONE ADMM iteration of the loop body src/solver.jl:140-165 including the
termination check every 25 iterations (the reference's own iter_time definition, src/solver.jl:134,169), with the
fixed-work settings of SURVEY 8d: eps_abs = eps_rel = 0, adaptive_rho_interval = 40, scaling = 10, alpha = 1.6,
sigma = 1e-6, rho = 0.1, CG tolerance 1/k^1.5, EmptyAccelerator.

Multi-GPU (--gpus N under torch.distributed.run): a single sparse QP does not shard ("replicas only"), so
every rank runs an independent replica of the workload on its own GPU; value = N*K / max-over-ranks time ("weak").

Output: ONE JSON line on rank 0 (see the driver contract), with `roofline` for the dominant kernel (the fused
[P | A'] operator SpMV of the CG iteration) and `cpu_baseline` (the CPU oracle timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_workload(args):
    import cosmo_jl_amd as cj
    if args.small:
        prob = cj.problems.sparse_box_qp(n=10_000, m=20_000, nnz=200_000)
    else:
        prob = cj.problems.sparse_box_qp()
    st = cj.Settings(eps_abs=0.0, eps_rel=0.0, max_iter=10 ** 9, check_infeasibility=10 ** 9, kkt_solver=cj.CGIndirectKKTSolver)
    return prob, st


def algorithmic_bytes(n, m, nnzA, nnzP):
    """SURVEY 8d per-launch compulsory bytes (fp64 values 8 B, int32 indices 4 B, every array once)."""
    b_A = 12.0 * nnzA + 4.0 * (m + 1) + 8.0 * n + 8.0 * m
    b_AT = 12.0 * nnzA + 4.0 * (n + 1) + 8.0 * m + 8.0 * n
    b_P = 12.0 * nnzP + 4.0 * (n + 1) + 16.0 * n
    # fused operator kernel c = [P | A'] [v; tmp] + sigma v: both matrices once, one row-pointer + one split array,
    # gather vectors v (n) and tmp (m) once, c written once, v re-read for the sigma term / dot product is cached
    b_op = 12.0 * (nnzA + nnzP) + 8.0 * (n + 1) + 8.0 * (n + m) + 8.0 * n
    b_vec = 8.0 * (8 * n + 13 * m)
    b_cgvec = 8.0 * (10 * n + m)
    return dict(A=b_A, AT=b_AT, P=b_P, op=b_op, vec=b_vec, cgvec=b_cgvec)


def cpu_baseline(prob, st_kwargs, sample_iters):
    """The CPU oracle ("port") on a bounded sample of the SAME workload: `sample_iters` ADMM iterations after setup, 1 thread.
    The loop runs in the compiled C restatement (oracle/cosmo_oracle_c.c, gcc -O2; Julia-style CSC SpMV kernels), set up by the
    NumPy oracle; if the compiled library is absent the NumPy/SciPy loop is timed instead and the sample string says so."""
    from oracle import cosmo_oracle as O
    from tests import util
    st = O.Settings(kkt_solver="cg", eps_abs=0.0, eps_rel=0.0, max_iter=sample_iters, check_infeasibility=10 ** 9)
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), st)
    try:
        from oracle import cosmo_oracle_c as OC
        OC.lib()
    except Exception:
        OC = None
    if OC is not None:
        c = OC.run(ws)
        iters, secs, cg = c["iter"], c["iter_time"], c["cg_iters_total"] / (c["iter"] + 1.0)
        impl = "compiled C loop (gcc -O2)"
    else:
        res = ws.optimize()
        iters, secs, cg = res.iter, res.iter_time, float(np.mean(res.cg_iters))
        impl = "NumPy/SciPy loop"
    return dict(value=iters / secs, unit="ADMM iterations/s", cores=1, kind="port",
                sample="%s: %d ADMM iterations (incl. init step, checks every 25) of the same cfg2 instance, %.1f s; mean CG its/solve %.2f"
                       % (impl, iters, secs, cg))


def max_over_ranks(elapsed, dist, device):
    """The bench contract's timing rule: every rank times its own K steps, the job's time is the MAX over the ranks."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_value(world, steps, elapsed):
    """Aggregate throughput of `world` replicas that each did `steps` iterations in `elapsed` seconds (weak scaling)."""
    return world * steps / elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--small", action="store_true", help="1/10-size instance (debugging only; not the BASELINE workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-iters", type=int, default=5)
    ap.add_argument("--exact-launches", action="store_true",
                    help="synchronise after every Krylov iteration (rocprofv3 runs: every launch does full work)")
    args = ap.parse_args()

    import torch
    import cosmo_jl_amd as cj

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)

    prob, st = build_workload(args)
    st.device = local_rank
    model = cj.Model()
    model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    cj.model.setup(model)                       # upload + Ruiz scaling on the device (setup!, excluded from iter_time)
    h = model.handle
    n, m = model.n, model.m
    nnzA, nnzP = model.A.nnz, model.P.nnz
    h.set_iterates(model.x, model.s, model.mu)
    h.admm_init()
    if args.exact_launches:
        h.set_profiling(2)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (untimed) ----
    h.admm_iterate_checked(args.warmup)
    stats0 = h.get_stats()
    barrier()
    t0 = time.perf_counter()
    h.admm_iterate_checked(args.steps)          # returns after the stream has drained (host sync inside)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        elapsed = max_over_ranks(elapsed, dist, "cuda")
        dist.barrier()
    stats1 = h.get_stats()
    steps_done = stats1["admm_iters"] - stats0["admm_iters"]
    assert steps_done == args.steps, (steps_done, args.steps)
    kbar = (stats1["kkt_iters_total"] - stats0["kkt_iters_total"]) / max(1, stats1["kkt_solves"] - stats0["kkt_solves"])
    value = whole_job_value(world, args.steps, elapsed)

    out = None
    if rank == 0:
        ab = algorithmic_bytes(n, m, nnzA, nnzP)
        # ---- dominant kernel: the fused [P | A'] operator SpMV of the CG iteration (K-bar launches per ADMM iteration).
        # Duration: HIP events on the library's stream around R back-to-back launches of exactly that kernel on the live
        # loop state (cosmo_hip_time_spmv; one event pair per R launches, so the ~5 us cost of an event pair does not
        # pollute a 10-20 us kernel; the ~1.6 us launch gap IS included, i.e. the number is conservative).
        t_op, bytes_op = h.time_spmv(cj._ffi.MAT_OP, 200)
        t_A, bytes_A = h.time_spmv(cj._ffi.MAT_A, 200)
        t_AT, bytes_AT = h.time_spmv(cj._ffi.MAT_AT, 200)
        t_P, bytes_P = h.time_spmv(cj._ffi.MAT_P, 200)
        achieved = bytes_op / t_op / 1e9
        # HBM traffic of one k_op_apply launch from the PMC counters: these need their own rocprofv3 --pmc passes (FETCH_SIZE
        # and WRITE_SIZE separately), so bench.py reports the committed result of the latest such pass over this very command
        traffic, traffic_src = None, None
        import glob
        for fpmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg2_pmc_traffic.json"))):
            try:
                with open(fpmc) as fh:
                    traffic = json.load(fh)["kernels"]["k_op_apply"]["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(fpmc, ROOT)
            except Exception:
                pass
        if args.small:
            traffic, traffic_src = None, None
        roof = dict(bound="hbm", kernel="k_op_apply (c = [P|A'][v; rho.*(A v)] + sigma v, CSR-stream SpMV)", achieved=round(achieved, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=bytes_op, avg_launch_us=round(t_op * 1e6, 3), launches_timed=200,
                    other={"k_spmv_A_rho (A v)": dict(achieved=round(bytes_A / t_A / 1e9, 1), frac=round(bytes_A / t_A / 1e9 / HBM_PEAK_GBS, 4),
                                                     algorithmic_bytes_per_launch=bytes_A, avg_launch_us=round(1e6 * t_A, 3)),
                           "k_cg_rhs (A' y)": dict(achieved=round(bytes_AT / t_AT / 1e9, 1), frac=round(bytes_AT / t_AT / 1e9 / HBM_PEAK_GBS, 4),
                                                  algorithmic_bytes_per_launch=bytes_AT, avg_launch_us=round(1e6 * t_AT, 3)),
                           "k_spmv_plain (P x)": dict(achieved=round(bytes_P / t_P / 1e9, 1), frac=round(bytes_P / t_P / 1e9 / HBM_PEAK_GBS, 4),
                                                     algorithmic_bytes_per_launch=bytes_P, avg_launch_us=round(1e6 * t_P, 3))})
        b_iter = ab["vec"] + (kbar + 2) * (ab["A"] + ab["AT"]) + (kbar + 1) * (ab["P"] + ab["cgvec"])
        out = {
            "metric": "ADMM iterations/sec (fp64) at fixed (n,m,nnz,cone)", "value": round(value, 3), "unit": "ADMM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: random sparse QP n=%d m=%d nnz(A)=%d nnz(P)=%d, Box cone, CG indirect KKT (tol 1/k^1.5), "
                                   "check_termination=25, adaptive_rho_interval=40, Ruiz scaling=10, eps=0" % (n, m, nnzA, nnzP),
                       "parallelism": "replicas x%d (a single sparse QP does not shard; SURVEY 8e)" % world,
                       "mean_cg_iters_per_admm_iter": round(kbar, 3), "kkt_budget_stalls": stats1["kkt_budget_stalls"],
                       "iteration_roofline_frac": round(b_iter * (value / world) / (HBM_PEAK_GBS * 1e9), 4),
                       "algorithmic_bytes_per_iteration": b_iter},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:     # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(prob, None, args.cpu_sample_iters if not args.small else 200)
            out["config"]["gpu_over_cpu"] = round((value / world) / out["cpu_baseline"]["value"], 2)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
