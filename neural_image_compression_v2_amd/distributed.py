"""Sample-sharded data parallelism for the fused training step.

The reference is single-device; its step is ``NUM_CROPS`` independent crops whose losses are averaged
(image_compression.py:233-265), so the path shards by crop: every rank holds the full grids + decoder, runs the
fused kernel on its crops with ``loss_scale = 1 / (3 * N_global)`` and ``sample_base`` = the global id of its first
sample (so the Philox noise does not depend on the world size), and ONE all-reduce(sum) of the flat buffer
``[loss | decoder grads | G0 grad | G1 grad]`` (fused.grad_bucket_layout) gives every rank the gradients of the
global batch.  One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests.

``step_fn`` is injected: the product passes fused.fused_forward_backward; tests pass the CPU oracle so the
sharding / bucket / scaling logic is exercised without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous, balanced split: (start, count); the first ``n_items % world`` ranks get one extra item"""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


@dataclass
class ShardPlan:
    rank: int
    world: int
    crop_start: int
    crop_count: int
    n_per_crop: int
    n_global: int

    @property
    def sample_base(self) -> int:
        return self.crop_start * self.n_per_crop

    @property
    def loss_scale(self) -> float:
        return 1.0 / (3.0 * self.n_global)


def plan_shard(num_crops_global: int, n_per_crop: int, rank: Optional[int] = None, world: Optional[int] = None) -> ShardPlan:
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    start, count = shard_range(num_crops_global, rank, world)
    return ShardPlan(rank, world, start, count, n_per_crop, num_crops_global * n_per_crop)


def all_reduce_flat(flat: torch.Tensor, group=None, async_op: bool = False):
    """the one exchange step of the path: sum of the flat gradient bucket over ranks"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def data_parallel_step(step_fn: Callable, geo, g0, g1, origins_global, params, target_global_or_local, *,
                       plan: Optional[ShardPlan] = None, group=None, target_is_local: bool = False, **step_kw):
    """Runs ``step_fn`` on this rank's crops and all-reduces the flat gradient bucket.

    geo            : fused.PathGeometry of the GLOBAL step (num_crops = global crop count)
    origins_global : [num_crops_global, dim] on the host (every rank draws the same seeded origins)
    target         : [N_global, 3] (sliced here) or, with target_is_local, this rank's [N_local, 3]
    returns the StepOutput of step_fn whose .flat now holds global sums (loss = global mean)
    """
    plan = plan or plan_shard(geo.num_crops, geo.n_per_crop)
    if plan.crop_count == 0:
        raise ValueError("more ranks than crops: give every rank at least one crop")
    org = torch.as_tensor(origins_global).reshape(-1, geo.dim)[plan.crop_start:plan.crop_start + plan.crop_count]
    tgt = target_global_or_local
    if not target_is_local:
        tgt = tgt.reshape(-1, 3)[plan.sample_base:plan.sample_base + plan.crop_count * plan.n_per_crop]
    local_geo = replace(geo, num_crops=plan.crop_count, sample_base=geo.sample_base + plan.sample_base, loss_scale=plan.loss_scale)
    out = step_fn(local_geo, g0, g1, org, params, tgt, **step_kw)
    all_reduce_flat(out.flat, group)
    return out
