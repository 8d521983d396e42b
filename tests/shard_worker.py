"""Worker process of tests/test_gpu_sharding.py: one RANK of a clique-sharded run (csrc/comm.hip) on device 0.

  python tests/shard_worker.py <transport> <rank> <world> <rendezvous> <out.npz> <iters>

transport = shm  : host-staged communicator (cosmo_hip_comm_init_hostshm); <rendezvous> is the POSIX shm name ("/cosmo_...")
transport = rccl : RCCL communicator; <rendezvous> is a file through which rank 0 hands the ncclUniqueId to the other ranks
                   (two ranks on ONE device: RCCL is expected to refuse this; the test records what it says)
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


def settings(iters):
    import cosmo_jl_amd as cj
    return cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)


def main():
    transport, rank, world, rdv, out, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    import cosmo_jl_amd as cj
    p = problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], settings(iters))
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
    bounds = cj.partition_cones_contiguous(cj.cone_costs(md.sets), world)
    h.set_cone_shard(bounds)
    r = cj.optimize(md)
    st = h.comm_stats()
    np.savez(out, x=r.x, s=r.s, y=r.y, iter=r.iter, kkt=r.kkt_iters_total, obj=r.obj_val, r_prim=r.info.r_prim, r_dual=r.info.r_dual,
             bounds=np.array(bounds), exchanges=st["exchanges"], nranks=st["nranks"], transport=st["transport"])


if __name__ == "__main__":
    main()
