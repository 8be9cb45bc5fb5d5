"""One rank of a multi-process sphx_multi run (started by tests/test_gpu_multi.py through torch.distributed.run): the halo records
travel through the caller-supplied communicator (torch.distributed over gloo — RCCL refuses two ranks on one GPU), the step loop
runs inside libsphx.  Writes this rank's owned particles to OUT/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, steps, scale, fixed_d, fixed_v = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch
    import torch.distributed as dist

    import yasph2d_amd as y
    from util import dam_break
    from yasph2d_amd.multi import MultiSolver, TorchCommOps

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    pos, boundary = dam_break(scale)
    comm = TorchCommOps(dist, torch.device("cuda", 0), shm_name="t" + os.environ.get("MASTER_PORT", "0"))
    m = MultiSolver.rank(y.default_params(fixed_iterations=(fixed_d, fixed_v)), 0, rank, world, comm=comm, halo=10, rebalance_every=4)
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    stats = [m.step(timer) for _ in range(steps)]
    d = m.download()
    info = m.info()
    np.savez(os.path.join(out, f"rank{rank}.npz"), pos=d["pos"], vel=d["vel"], density=d["density"], ids=d["ids"],
             iters=np.array([(s["density_iterations"], s["divergence_iterations"]) for s in stats]), dt_ns=np.array([s["dt_ns"] for s in stats]),
             exchanges=info["exchanges"], rebalances=info["rebalances"])
    m.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
