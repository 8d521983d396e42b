"""Worker process of tests/test_gpu_sharding.py: one RANK of a clique-sharded run (csrc/comm.hip) on device 0.

  python tests/shard_worker.py <transport> <rank> <world> <rendezvous> <out.npz> <iters>

transport = shm  : host-staged communicator (cosmo_hip_comm_init_hostshm); <rendezvous> is the POSIX shm name ("/cosmo_...")
transport = rccl : RCCL communicator; <rendezvous> is a file through which rank 0 hands the ncclUniqueId to the other ranks
                   (two ranks on ONE device: RCCL is expected to refuse this; the test records what it says)
Environment: COSMO_TEST_SHARD = cones (default: cosmo_hip_set_cone_shard, the projections only) | rows (cosmo_hip_set_row_shard: cones +
their rows, csrc/rowshard.hip); COSMO_TEST_CASE = chordal (default) | pinf | dinf (the two infeasible problems of
test_infeasibility_certificates_in_sharded_runs, default settings); COSMO_TEST_TIGHT=1: CG solved to 1e-10 (tol_exponent 0);
COSMO_TEST_DTYPE=float32: the Float32 library; COSMO_TEST_ACCEL=1: the reference's default accelerator (COSMO_TEST_ACCEL_VARIANT=type1_rolling: the Type1 / RollingMemory variant) (AndersonAccelerator, mem 15, safeguarded) with a
tight CG and eps = 1e-6 -- a convergent accelerated run (`iters` is then max_iter).
Writes the final iterates, the result scalars and the communicator statistics of this rank."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def problem():
    import cosmo_jl_amd as cj
    # cliques of both tile classes of the batched matrix-sign path (single 64 x 64 tile, d <= 64, and multi-tile) + SOC-free simple rows
    return cj.problems.chordal_sdp(ncliques=14, dmin=6, dmax=110, sep_min=1, sep_max=4, n_total=3000, n_zero=20, n_nonneg=40, seed=55)


def settings(iters, tight=False):
    import cosmo_jl_amd as cj
    if os.environ.get("COSMO_TEST_ACCEL", "") == "1":
        acc = cj.AndersonAccelerator
        if os.environ.get("COSMO_TEST_ACCEL_VARIANT", "") == "type1_rolling":        # docs/src/acceleration.md:23
            acc = cj.AndersonAccelerator[cj.Type1, cj.RollingMemory]
        return cj.Settings(max_iter=iters, eps_abs=1e-6, eps_rel=1e-6, accelerator=acc,
                           kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0))
    kw = dict(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    if os.environ.get("COSMO_TEST_AUTO_RHO"):               # the automatic rho interval (adaptive_rho_interval = 0, solver.jl:244-256); setup times per rank: see main()
        kw["adaptive_rho_interval"] = 0
    if os.environ.get("COSMO_TEST_TIMELIMIT"):              # a wall-clock limit (solver.jl:351-354): the ranks must stop at the same iteration
        kw["time_limit"] = float(os.environ["COSMO_TEST_TIMELIMIT"])
    if tight:
        kw["kkt_solver"] = cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)
    if os.environ.get("COSMO_TEST_KKT", "") == "minres_reduced":      # IndirectReducedKKTSolver with solver_type = :MINRES (kktsolver_indirect.jl:3-88), solved tightly
        kw["kkt_solver"] = cj.with_options(cj.IndirectReducedKKTSolverMINRES, tol_constant=1e-10, tol_exponent=0.0)
    if os.environ.get("COSMO_TEST_KKT", "") == "cg_jacobi":           # the opt-in Jacobi-preconditioned CG (kkt_kind CG_JACOBI) on the assembled reduced operator
        kw["kkt_solver"] = cj.with_options(cj.CGJacobiKKTSolver, tol_constant=1e-10, tol_exponent=0.0)
    return cj.Settings(**kw)


def infeasible_model(case):
    """pinf: X psd and X11 = -1 (3 x 3, svec variables) + an SOC block; dinf: minimise -t over a second-order cone, + a free PSD block
    so that two ranks own cones.  Default settings (certificates every 40 iterations)."""
    import scipy.sparse as sp
    import cosmo_jl_amd as cj
    md = cj.Model()
    if case == "pinf":
        nt = 6
        A = sp.vstack([sp.csc_matrix(([1.0], ([0], [0])), shape=(1, nt + 3)), sp.hstack([sp.identity(nt), sp.csc_matrix((nt, 3))]),
                       sp.hstack([sp.csc_matrix((3, nt)), sp.identity(3)])], format="csc")
        b = np.concatenate([[1.0], np.zeros(nt + 3)])
        cons = [cj.Constraint(A[:1], b[:1], cj.ZeroSet), cj.Constraint(A[1:1 + nt], b[1:1 + nt], cj.PsdConeTriangle),
                cj.Constraint(A[1 + nt:], b[1 + nt:], cj.SecondOrderCone)]
        cj.assemble(md, sp.csc_matrix((nt + 3, nt + 3)), np.zeros(nt + 3), cons, settings=cj.Settings(kkt_solver=cj.CGIndirectKKTSolver))
    else:
        A = sp.identity(9, format="csc")
        cons = [cj.Constraint(A[:3], np.zeros(3), cj.SecondOrderCone), cj.Constraint(A[3:], np.zeros(6), cj.PsdConeTriangle)]
        q = np.zeros(9); q[0] = -1.0
        cj.assemble(md, sp.identity(9, format="csc") * 0.0, q, cons, settings=cj.Settings(kkt_solver=cj.CGIndirectKKTSolver))
    return md


def custom_cone_model(iters):
    """The reference's worked example of a user-defined cone (docs/src/literate/custom_cone.jl:19-49: Nonpositives as an AbstractConvexCone plugin,
    LP with solution x = (3, 2, 2)) next to a decoupled 3 x 3 PSD block, so that two ranks own cones and ONE of them owns the custom cone."""
    import scipy.sparse as sp
    import cosmo_jl_amd as cj

    class Nonpositives(cj.AbstractConvexCone):
        def project(self, x):
            np.minimum(x, 0.0, out=x)

    nz = 6
    Z0 = np.array([[1.0, 2.0, 0.0], [2.0, -3.0, 0.5], [0.0, 0.5, 0.2]])
    P = sp.block_diag([sp.csc_matrix((3, 3)), sp.identity(nz)], format="csc")
    q = np.concatenate([-np.ones(3), -cj.problems.svec(Z0)])
    A1 = sp.hstack([sp.csc_matrix(np.array([[1.0, 0, 0], [0, 1.0, 0]])), sp.csc_matrix((2, nz))], format="csc"); b1 = np.array([-3.0, -2.0])
    A2 = sp.hstack([sp.csc_matrix(np.array([[1.0, 0, 1.0]])), sp.csc_matrix((1, nz))], format="csc"); b2 = np.array([-5.0])
    A3 = sp.hstack([sp.csc_matrix((nz, 3)), sp.identity(nz)], format="csc"); b3 = np.zeros(nz)
    md = cj.Model()
    cj.assemble(md, P, q, [cj.Constraint(A1, b1, Nonpositives), cj.Constraint(A2, b2, cj.ZeroSet), cj.Constraint(A3, b3, cj.PsdConeTriangle)],
                settings=cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9,
                                     kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)))
    return md


def build_model(iters):
    import cosmo_jl_amd as cj
    case = os.environ.get("COSMO_TEST_CASE", "chordal")
    if case in ("pinf", "dinf"):
        return infeasible_model(case)
    if case == "custom":
        return custom_cone_model(iters)
    if case == "nanrow":          # tests/test_gpu_nan_residuals.py: an empty row of A with b = NaN, owned by rank 1 (local row 100: upper half of wave 1)
        from tests import util
        p = util.nan_row_qp(3, 60, 2000, 1300, split=1200)
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"],
                                cj.Settings(max_iter=iters, scaling=0, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)))
        return md
    p = problem()
    if case == "chordal_split":
        # the same problem with its ZeroSet(20) written as ZeroSet(1) + ZeroSet(19): a cone of ONE row, so that an explicit partition (COSMO_TEST_BOUNDS)
        # can give a rank a single row and no PSD cone; rows, matrices and therefore the single-rank iterates are unchanged
        z = p["sets"][0]
        assert z.kind == cj._ffi.ZERO and z.dim == 20
        p["sets"] = [cj.ZeroSet(1), cj.ZeroSet(19)] + list(p["sets"][1:])
    dtype = np.float32 if os.environ.get("COSMO_TEST_DTYPE", "") == "float32" else np.float64       # libcosmo_hip_f32.so: the collectives carry float
    md = cj.Model(dtype=dtype); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], settings(iters, os.environ.get("COSMO_TEST_TIGHT", "") == "1"))
    return md


def main():
    transport, rank, world, rdv, out, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    import cosmo_jl_amd as cj
    md = build_model(iters)
    cj.model.setup(md)
    h = md.handle
    if transport == "shm":
        h.comm_init_hostshm(rank, world, rdv)
    else:
        if rank == 0:
            uid = cj.Handle.comm_unique_id()
            with open(rdv + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(rdv + ".tmp", rdv)
        else:
            t0 = time.time()
            while not os.path.exists(rdv):
                if time.time() - t0 > 60:
                    raise SystemExit("no unique id from rank 0")
                time.sleep(0.05)
            uid = open(rdv, "rb").read()
        h.comm_init(rank, world, uid)
    mode = os.environ.get("COSMO_TEST_SHARD", "cones")
    explicit = os.environ.get("COSMO_TEST_BOUNDS")              # "0,1,3,...": first cone of every rank + the number of cones (an explicit partition)
    if mode == "rows":
        bounds = [int(v) for v in explicit.split(",")] if explicit else cj.partition_cones_contiguous(cj.model.row_shard_costs(md.sets), world)
        h.set_row_shard(bounds)
    else:
        bounds = [int(v) for v in explicit.split(",")] if explicit else cj.partition_cones_contiguous(cj.cone_costs(md.sets), world)
        h.set_cone_shard(bounds)
    if os.environ.get("COSMO_TEST_AUTO_RHO"):
        # every rank is GIVEN its own ws.times.setup_time ("a,b,..." seconds by rank; model.optimize() would hand over this process's own measurement)
        forced = float(os.environ["COSMO_TEST_AUTO_RHO"].split(",")[rank])
        orig = h.set_setup_time
        h.set_setup_time = lambda t: orig(forced)
    r = cj.optimize(md)
    st = h.comm_stats()
    ex = h.comm_stats_ex()
    info = h.row_shard_info()
    acc = h.accel_stats() if os.environ.get("COSMO_TEST_ACCEL", "") == "1" else dict(accelerated=0, accepted=0, declined=0, safeguarding_iter=0)
    np.savez(out, accelerated=acc["accelerated"], declined=acc["declined"], safeguarding_iter=acc["safeguarding_iter"], x=r.x, s=r.s, y=r.y, iter=r.iter, kkt=r.kkt_iters_total, obj=r.obj_val, r_prim=r.info.r_prim, r_dual=r.info.r_dual,
             bounds=np.array(bounds), exchanges=st["exchanges"], nranks=st["nranks"], transport=st["transport"], status=r.status,
             mode=ex["mode"], bytes=ex["bytes"], allreduces=ex["allreduces"], allreduce_elems=ex["allreduce_elems"],
             row_lo=info["row_lo"], row_hi=info["row_hi"], rho_updates=np.array(r.info.rho_updates), rho_interval=np.array(h.rho_interval()))


if __name__ == "__main__":
    main()
