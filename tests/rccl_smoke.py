"""Two ranks, one GPU each, RCCL (backend "nccl"): the collectives of the sharded path with UNEVEN shards --
DistBatchReducer.device_any / device_all / decide / mean_abs and gather_solution.  Launched by tests/test_gpu_rccl.py under
torch.distributed.run (needs >= 2 GPUs); prints "RCCL-SMOKE-OK" from rank 0."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("THX_SMOKE_BACKEND") == "gloo":   # the same script on CPU (tests/test_sharded_lm.py): logic check only
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
    else:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    from theseus_amd.sharding import DistBatchReducer, gather_solution, shard_bounds
    red = DistBatchReducer()
    assert red.world_size == world
    # a flag raised on the LAST rank only is seen by everybody, on the device (no host sync inside)
    flag = torch.tensor(rank == world - 1, device=dev)
    assert bool(red.device_any(flag)) and not bool(red.device_all(flag))
    assert bool(red.device_all(torch.tensor(True, device=dev)))
    any_r, all_r = red.decide([torch.tensor([rank == 0], device=dev)], [torch.tensor([rank != 1], device=dev)])
    assert any_r == [True] and all_r == [False]
    # uneven shards of a batch of 7 problems, P = 3 poses
    total, P = 7, 3
    lo, hi = shard_bounds(total, rank, world)
    full = torch.arange(P * total * 12, dtype=torch.float64, device=dev).view(P, total, 3, 4)
    got = gather_solution(full[:, lo:hi].contiguous())
    assert got.shape == full.shape and torch.equal(got, full), (got.shape, full.shape)
    err = torch.arange(lo, hi, dtype=torch.float64, device=dev)
    assert abs(red.mean_abs(err) - (total - 1) / 2.0) < 1e-12
    assert bool(red.device_mean_abs_below(err, 3.5)) and not bool(red.device_mean_abs_below(err, 2.5))
    dist.barrier()
    if rank == 0:
        print("RCCL-SMOKE-OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
