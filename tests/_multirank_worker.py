"""Rank process of tests/test_gpu_multirank.py: one rank of a world_size-N job sharing the box's one GPU (gloo for the
collectives, every rank on cuda:0), running the REAL HIP rollout through parallel.sharded_rollout.

    python _multirank_worker.py <rank> <world> <port> <out.npz>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.parallel import sharded_rollout, broadcast_kinv
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O              # checker-side: the CPU inverse every rank must receive bit-identically

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from datetime import timedelta
    backend = os.environ.get("GPMPC_TEST_BACKEND", "gloo")        # "nccl" (= RCCL): one rank per GPU
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index),
                                timeout=timedelta(seconds=120))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    pb = synth_problem(23, 140, 3, 1, 5, 8)
    # the inverse comes from rank 0 only (replicas are built from the same bits)
    if rank == 0:
        kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.clone()
    else:
        kinv = torch.zeros((3, 140, 140), dtype=torch.float64)
    if backend == "nccl":                                         # RCCL moves device tensors
        kinv = kinv.cuda()
    broadcast_kinv(kinv, dist, src=0)
    kinv = kinv.cpu()
    pack = g.GPPack(pb["X"], pb["Y"], kinv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = g.CostParams(-1.0, pb["Q"], pb["R"])
    fn = lambda x0b, Ub: g.rollout(pack, x0b, Ub, cost, want_traj=False)      # noqa: E731
    res = {}
    for tag, B in (("even", 8), ("ragged", 7), ("fewer_than_ranks", 1)):
        x0 = torch.as_tensor(pb["x0"][:B], device="cuda")
        U = torch.as_tensor(pb["U"][:B], device="cuda")
        c, gr = sharded_rollout(fn, x0, U, dist)
        res[tag + "_cost"], res[tag + "_grad"] = c.cpu().numpy(), gr.cpu().numpy()
    if rank == 0:
        np.savez(out, kinv=kinv.numpy(), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
