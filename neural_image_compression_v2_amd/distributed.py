"""Sample-sharded data parallelism for the fused training step.

The reference is single-device; its step is ``NUM_CROPS`` independent crops whose losses are averaged
(image_compression.py:233-265), so the path shards by crop: every rank holds the full grids + decoder, runs the
fused kernel on its crops with ``loss_scale = 1 / (3 * N_global)`` and ``sample_base`` = the global id of its first
sample (so the in-kernel Threefry-4x32-12 noise does not depend on the world size), and ONE all-reduce(sum) of the flat buffer
``[loss | decoder grads | G0 grad | G1 grad]`` (fused.grad_bucket_layout) gives every rank the gradients of the
global batch.  One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests.

``step_fn`` is injected: the product passes fused.fused_forward_backward; tests pass the CPU oracle so the
sharding / bucket / scaling logic is exercised without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
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
    if flat.is_cuda and dist.get_backend(group) == "gloo":             # one-GPU rehearsals: through the host
        return _all_reduce_sum(flat, group)
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def data_parallel_step(step_fn: Callable, geo, g0, g1, origins_global, params, target_global_or_local, *,
                       plan: Optional[ShardPlan] = None, group=None, target_is_local: bool = False, **step_kw):
    """Runs ``step_fn`` on this rank's crops and all-reduces the flat gradient bucket.

    geo            : fused.PathGeometry of the GLOBAL step (num_crops = global crop count)
    origins_global : [num_crops_global, dim] on the host (every rank draws the same seeded origins)
    target         : [N_global, 3] (sliced here) or, with target_is_local, this rank's [N_local, 3]
    returns the StepOutput of step_fn whose .flat now holds global sums (loss = global mean)
    """
    if int(getattr(geo, "passes", 1)) != 1:
        # sample ids are (crop * passes + pass) * n_per_crop + ...: the crop shards of this function assume one pass per crop
        # (n_global, sample_base and the target slice would all be off by the factor `passes`); repeated passes belong to the
        # stripe-sharded step (StripePlan), which takes them from one crop
        raise ValueError("data_parallel_step shards crops with passes == 1; use the stripe-sharded step for repeated passes")
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


# ------------------------------------------------------------------------------------------------------------------------------
# Stripe-sharded grids: the exchange shrinks from the whole gradient bucket to one node row per stripe boundary
# ------------------------------------------------------------------------------------------------------------------------------
@dataclass
class StripePlan:
    """Whole-image passes (every sample of the image/volume once per rank and step): instead of replicating the grids and
    all-reducing their dense gradients (4K image: 31 MB per step over point-to-point xGMI links), every rank owns a STRIPE of the
    last sample axis - the slowest spatial axis of the grid tensors ``[C, (Z,) Y, X]`` (fp_def.py:54,76), so a stripe is one
    contiguous block of node rows per channel - and draws all its ``world`` crops per step from that stripe.  The sample multiset
    of a step is the same as with replicated grids (every sample ``world`` times, distinct noise ids); a rank reads and updates
    only the nodes of its stripe, so the only grid gradients two ranks share are the ONE node row of G0 and of G1 on each stripe
    boundary.  ``exchange`` sums those rows together with the loss and the decoder gradients in one small all-reduce;
    ``assemble`` rebuilds the full grids from the stripes once, after training.

    Stripes start at multiples of the G1 cell (``g1_cell`` samples), so a boundary is a node row of both grids.
    """
    rank: int
    world: int
    length: int          # samples along the stripe axis
    g1_cell: int         # samples per G1 cell along it (2 x the G0 cell)
    starts: List[int]    # stripe starts (samples), len world + 1; starts[-1] = length
    _index: dict = field(default_factory=dict, repr=False, compare=False)

    @property
    def start(self) -> int:
        return self.starts[self.rank]

    @property
    def size(self) -> int:
        return self.starts[self.rank + 1] - self.starts[self.rank]

    def node_rows(self, level: int, rank: Optional[int] = None) -> Tuple[int, int]:
        """first and last node row (inclusive) of grid ``level`` (0: G0, 1: G1) that the stripe's samples touch"""
        r = self.rank if rank is None else rank
        cell = self.g1_cell // 2 if level == 0 else self.g1_cell
        return self.starts[r] // cell, -(-self.starts[r + 1] // cell)

    def boundary_rows(self, level: int) -> List[int]:
        cell = self.g1_cell // 2 if level == 0 else self.g1_cell
        return [s // cell for s in self.starts[1:-1]]

    def boundary_index(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """the boundary rows of G0 and G1 as index tensors on ``device`` (built once per device)"""
        key = str(device)
        if key not in self._index:
            self._index[key] = tuple(torch.tensor(self.boundary_rows(level), dtype=torch.long, device=device) for level in (0, 1))
        return self._index[key]


def plan_stripes(length: int, g1_cell: int, rank: Optional[int] = None, world: Optional[int] = None) -> StripePlan:
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if g1_cell < 2 or g1_cell % 2:
        raise ValueError("g1_cell must be an even number of samples (2 x the G0 cell)")
    cells = length // g1_cell
    if cells < world:
        raise ValueError(f"{length} samples hold {cells} G1 cells: fewer than {world} ranks")
    starts = [shard_range(cells, r, world)[0] * g1_cell for r in range(world)] + [int(length)]
    return StripePlan(int(rank), int(world), int(length), int(g1_cell), starts)


def _all_reduce_sum(t: torch.Tensor, group=None) -> None:
    """in-place sum over ranks; device tensors go through the host when the group's backend is gloo (one-GPU rehearsals, tests)"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _row_set(plan: StripePlan, level: int, grad: torch.Tensor):
    """nic_row_set of a grid-gradient tensor [C, rows, ...]: the stripe axis is tensor axis 1, a node row = everything behind it"""
    from . import _lib
    rows = plan.boundary_rows(level)
    if len(rows) > _lib.NIC_STRIPE_MAX_ROWS:
        raise ValueError(f"{len(rows)} stripe boundaries: the exchange kernels take at most {_lib.NIC_STRIPE_MAX_ROWS}")
    rs = _lib.NicRowSet()
    rs.base = grad.data_ptr()
    rs.plane = int(grad[0].numel())
    rs.row_elems = int(grad[0, 0].numel())
    rs.channels = int(grad.shape[0])
    rs.nrows = len(rows)
    for i, r in enumerate(rows):
        rs.rows[i] = int(r)
    return rs


def stripe_exchange(plan: StripePlan, small: torch.Tensor, grad_g0: torch.Tensor, grad_g1: torch.Tensor, group=None,
                    reduce: Optional[Callable] = None, overlap: Optional[Callable] = None) -> None:
    """The exchange step of a stripe-sharded training step, in place: ``small`` (loss + decoder gradients, a 1-D view of the flat
    bucket) and the boundary node rows of the two grid gradients become sums over ranks - ONE all-reduce of
    ``small.numel() + (world - 1) * C * (row of G0 + row of G1)`` floats.  On the device the buffer is packed and unpacked by one
    launch each (``nic_stripe_pack`` / ``nic_stripe_unpack``); CPU tensors (the gloo tests with the oracle as step function) take the
    torch formulation below.

    ``overlap``: work that needs NOTHING the exchange produces - the optimiser update of the rank's interior node rows
    (``stripe_row_parts``) - called once the collective has been started and before its results are written back: on the device the
    all-reduce is issued asynchronously (RCCL runs it on its own stream), ``overlap()`` launches onto the caller's stream meanwhile, and the
    caller's stream only waits for the collective before the unpack.  The boundary rows and the decoder are updated by the caller afterwards."""
    if plan.world == 1:
        if overlap is not None:
            overlap()
        return
    if grad_g0.is_cuda and grad_g0.is_contiguous() and grad_g1.is_contiguous() and small.is_contiguous():
        import ctypes
        from . import _lib
        lib = _lib.load()
        # the buffer is sized from `small` and the grids' shapes: both are part of the key (a decoder of another depth, or a reallocated gradient
        # of another shape at the same address, must not reuse a buffer that is too short - the C side cannot check its length)
        key = ("xbuf", str(grad_g0.device), grad_g0.data_ptr(), grad_g1.data_ptr(), small.numel(), tuple(grad_g0.shape), tuple(grad_g1.shape))
        cached = plan._index.get(key)
        if cached is None:
            sets = (_lib.NicRowSet * 2)(_row_set(plan, 0, grad_g0), _row_set(plan, 1, grad_g1))
            n = small.numel() + sum(int(q.channels) * int(q.nrows) * int(q.row_elems) for q in sets)
            cached = plan._index[key] = (sets, torch.empty(n, dtype=torch.float32, device=grad_g0.device))
        sets, buf = cached
        assert buf.numel() == small.numel() + sum(int(q.channels) * int(q.nrows) * int(q.row_elems) for q in sets)
        with torch.cuda.device(grad_g0.device):
            st = _lib.stream_ptr(grad_g0.device)
            _lib.check(lib.nic_stripe_pack(_lib.ptr(small), small.numel(), sets, 2, _lib.ptr(buf), st), "nic_stripe_pack")
            work = None
            if reduce is not None:
                reduce(buf, group)
            elif dist.get_backend(group) == "gloo":
                _all_reduce_sum(buf, group)                               # through the host (one-GPU rehearsals): nothing to overlap with
            else:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
            if overlap is not None:
                overlap()                                                 # interior rows: under the collective
            if work is not None:
                work.wait()                                               # the current stream waits for RCCL's; the host does not block
            _lib.check(lib.nic_stripe_unpack(_lib.ptr(small), small.numel(), sets, 2, _lib.ptr(buf), st), "nic_stripe_unpack")
        return
    idx0, idx1 = plan.boundary_index(grad_g0.device)
    h0, h1 = grad_g0.index_select(1, idx0), grad_g1.index_select(1, idx1)
    ns, n0 = small.numel(), h0.numel()
    buf = torch.cat([small.reshape(-1), h0.reshape(-1), h1.reshape(-1)])
    (reduce or _all_reduce_sum)(buf, group)                                 # `reduce`: test / rehearsal hook
    if overlap is not None:
        overlap()                                                           # before the sums are written back: it must not need them
    small.copy_(buf[:ns].view_as(small))
    grad_g0.index_copy_(1, idx0, buf[ns:ns + n0].view_as(h0))
    grad_g1.index_copy_(1, idx1, buf[ns + n0:].view_as(h1))


def stripe_param_blocks(plan: StripePlan, level: int, *tensors: torch.Tensor) -> List[Tuple[torch.Tensor, ...]]:
    """Optimiser sharding: the node rows of grid ``level`` this rank's samples touch (its stripe incl. the boundary row on either side),
    as one contiguous block per channel of every given tensor of the grid's shape ``[C, rows, ...]`` (parameter, gradient, 16-bit
    mirror).  A rank runs Adam over these blocks ONLY - 1 / world of the optimiser work, moments allocated for them only
    (``stripe_state``) - so nothing outside its stripe ever moves, whatever state a resumed run starts from (round 2 ran Adam over
    the whole tensors and relied on zero gradients AND zero moments outside the stripe)."""
    lo, hi = plan.node_rows(level)
    return [tuple(t[c, lo:hi + 1] for t in tensors) for c in range(tensors[0].shape[0])]


def stripe_row_parts(plan: StripePlan, level: int) -> Tuple[Tuple[int, int], List[int]]:
    """The rank's node rows of grid ``level`` split for the overlapped step: ``((lo, hi), boundary)`` - the inclusive range of INTERIOR rows
    (touched by this rank's samples only: their gradients are final when the fused kernel ends; ``lo > hi`` when there are none) and the
    rank's boundary rows (shared with a neighbour: final after the exchange) - at most two, the first and / or last row of its stripe."""
    lo, hi = plan.node_rows(level)
    shared = set(plan.boundary_rows(level))
    boundary = [r for r in (lo, hi) if r in shared]
    if lo == hi:
        boundary = boundary[:1]
    return (lo + (1 if lo in shared else 0), hi - (1 if hi in shared else 0)), boundary


def stripe_state(plan: StripePlan, level: int, grid: torch.Tensor) -> torch.Tensor:
    """zero Adam moments for this rank's node rows of ``grid``: ``[C, own rows, ...]`` (channel c = the block of stripe_param_blocks)"""
    lo, hi = plan.node_rows(level)
    return torch.zeros((grid.shape[0], hi - lo + 1) + tuple(grid.shape[2:]), dtype=torch.float32, device=grid.device)


def assemble_stripes(plan: StripePlan, g0: torch.Tensor, g1: torch.Tensor, group=None) -> None:
    """After training: every rank keeps the node rows it owns (from its first row up to, not including, the next stripe's first
    row; the last rank up to the end of the grid), zeroes the rest, and one all-reduce(sum) gives everyone the full grids.
    Boundary rows are identical on both neighbours (same summed gradients, same optimiser state), so either copy would do."""
    if plan.world == 1:
        return
    for level, g in ((0, g0), (1, g1)):
        lo = plan.node_rows(level)[0]
        hi = plan.node_rows(level, plan.rank + 1)[0] if plan.rank + 1 < plan.world else g.shape[1]
        keep = g[:, lo:hi].clone()
        g.zero_()
        g[:, lo:hi] = keep
        _all_reduce_sum(g, group)
