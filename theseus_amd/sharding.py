"""Batch sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards along the batch dimension: problems are independent, every kernel is batch-leading and
no kernel mixes batch elements, so rank r simply owns a contiguous slice of the batch and
there is NO collective on the data path.  What does cross shards:

* three batch-global predicates of the reference loop -- ``err.abs().mean() < abs_err_tolerance``
  (nonlinear_optimizer.py:111), ``converged.all()`` (nonlinear_least_squares.py:202) and
  ``reject.all()`` / ``reject.any()`` (:358,362) -- plus "some linear solve failed", which stops the
  whole batch (:138-152).  ``DistBatchReducer`` turns them into one small all-reduce per iteration so that
  a sharded run takes exactly the control path of the unsharded reference run;
* the solution: one ``all_gather`` of the final poses (``gather_solution``), if the caller wants it
  everywhere.
"""
from typing import List, Sequence, Tuple

import torch


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``total`` problems: the first ``total % world`` ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def plan_sub_batches(total: int, rank: int, world: int, max_batch: int) -> Tuple[int, int]:
    """Strong scaling of a job of ``total`` problems (BASELINE.json configs[2]): this rank's share (``shard_bounds``) is
    solved as ``n_sub`` consecutive sub-batches of ``batch`` problems through one set of persistent workspaces.
    Returns (batch, n_sub); the share must be at most ``max_batch`` or a multiple of it."""
    lo, hi = shard_bounds(total, rank, world)
    share = hi - lo
    if share <= 0:
        raise ValueError(f"rank {rank} of {world} has no problems in a job of {total}")
    if share <= max_batch:
        return share, 1
    if share % max_batch:
        raise ValueError(f"rank share {share} is not a multiple of the sub-batch {max_batch}")
    return max_batch, share // max_batch


def shard_tensors(tensors, rank: int, world: int):
    """name -> tensor dict with batch-leading tensors -> this rank's slice (batch-1 tensors are shared)."""
    B = max(t.shape[0] for t in tensors.values())
    lo, hi = shard_bounds(B, rank, world)
    return {k: (t if t.shape[0] == 1 else t[lo:hi].contiguous()) for k, t in tensors.items()}


class LocalBatchReducer:
    """Single process: the predicates are evaluated on the local batch, one host sync."""

    world_size = 1

    def decide(self, any_flags: Sequence[torch.Tensor], all_flags: Sequence[torch.Tensor]) -> Tuple[List[bool], List[bool]]:
        """OR-reduce every tensor of ``any_flags`` and AND-reduce every tensor of ``all_flags`` over the whole
        (global) batch; returns python bools.  One device->host transfer."""
        parts = [f.any().view(1) for f in any_flags] + [(~f.bool()).any().view(1) for f in all_flags]
        if not parts:
            return [], []
        v = self._max(torch.cat(parts).to(torch.int32)).tolist()
        na = len(any_flags)
        return [bool(x) for x in v[:na]], [not bool(x) for x in v[na:]]

    def mean_abs(self, err: torch.Tensor) -> float:
        s = self._sum(torch.stack([err.abs().sum().double(), torch.tensor(float(err.numel()), dtype=torch.float64,
                                                                          device=err.device)]))
        s = s.tolist()
        return s[0] / s[1]

    def device_any(self, flag: torch.Tensor) -> torch.Tensor:
        """OR of a 0-dim bool tensor over the global batch, result stays on the device (no host sync)."""
        return flag

    def device_all(self, flag: torch.Tensor) -> torch.Tensor:
        """AND of a 0-dim bool tensor over the global batch, on the device."""
        return ~self.device_any(~flag)

    def device_mean_abs_below(self, err: torch.Tensor, tol: float) -> torch.Tensor:
        """0-dim bool on the device: mean(|err|) over the GLOBAL batch < tol (nonlinear_optimizer.py:111)."""
        s = self._sum_device(torch.stack([err.abs().sum().double(),
                                          torch.tensor(float(err.numel()), dtype=torch.float64, device=err.device)]))
        return (s[0] / s[1]) < tol

    def _sum_device(self, t):
        return t

    def _max(self, t):
        return t

    def _sum(self, t):
        return t


class DistBatchReducer(LocalBatchReducer):
    """One process per GPU: the same predicates over the union of all shards (all-reduce MAX / SUM of a
    handful of scalars per LM iteration)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("DistBatchReducer needs an initialised torch.distributed process group")
        self.dist, self.group = dist, group
        self.world_size = dist.get_world_size(group)

    def device_any(self, flag):
        t = flag.to(torch.int32).view(1)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)   # stream-ordered: the host does not wait
        return t.bool().view(())

    def _sum_device(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)   # stream-ordered, no host wait
        return t

    def _max(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def _sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


def gather_solution(poses: torch.Tensor, group=None) -> torch.Tensor:
    """(P, B_local, 3, 4) on every rank -> (P, B_total, 3, 4) on every rank, shards in rank order: the one
    collective of the data path.  Uneven shards (shard_bounds) are padded to the largest for the exchange."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=poses.device)
    sizes[dist.get_rank(group)] = poses.shape[1]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = sizes.tolist()
    bmax = max(sizes)
    local = poses.contiguous()
    if local.shape[1] != bmax:
        pad = torch.zeros((poses.shape[0], bmax - poses.shape[1]) + tuple(poses.shape[2:]), dtype=poses.dtype,
                          device=poses.device)
        local = torch.cat([local, pad], 1)
    out = torch.empty((world * poses.shape[0], bmax) + tuple(poses.shape[2:]), dtype=poses.dtype, device=poses.device)
    dist.all_gather_into_tensor(out, local, group=group)  # concatenated form: accepted by RCCL and gloo alike
    out = out.view((world, poses.shape[0], bmax) + tuple(poses.shape[2:]))
    return torch.cat([out[r, :, :sizes[r]] for r in range(world)], 1)
