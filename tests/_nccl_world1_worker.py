"""Rank process of tests/test_gpu_multirank.py::test_rccl_backend_code_path_world1: the RCCL ("nccl") branch of
parallel.gather_results / broadcast_kinv and of bench.py's rank set-up, on the one GPU of the box (world_size 1)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = sys.argv[1]
    import torch.distributed as dist
    from gaussian_process_mpc_amd.parallel import gather_results, broadcast_kinv, sharded_rollout
    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    pb = synth_problem(29, 100, 2, 1, 4, 6)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.to(dev)
    broadcast_kinv(kinv, dist, src=0)
    pack = g.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"], device=dev)
    cost = g.CostParams(-1.0, pb["Q"], pb["R"])
    x0, U = torch.as_tensor(pb["x0"], device=dev), torch.as_tensor(pb["U"], device=dev)
    r = g.rollout(pack, x0, U, cost, want_traj=False)
    c, gr = gather_results(r["cost"], r["grad"], dist)                     # all_gather_into_tensor over RCCL
    assert torch.equal(c, r["cost"]) and torch.equal(gr, r["grad"])
    c2, g2 = sharded_rollout(lambda a, b: g.rollout(pack, a, b, cost, want_traj=False), x0, U, dist)
    assert torch.equal(c2, r["cost"]) and torch.equal(g2, r["grad"])
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl world-1 ok")


if __name__ == "__main__":
    main()
