"""Multi-GPU layout of the attack path: one process per GPU, independent attack instances per rank.

The path shards by sample (every quantity in the loop is per-sample), so there is NO per-iteration
collective. RCCL (torch.distributed backend "nccl") is used exactly twice per job:
  * broadcast_frozen_weights — ONE broadcast of the flattened fp32 parameter+buffer blob of the frozen victim(s)
    (PointNet ~ 6.5 MB) from rank 0 over xGMI at start-up;
  * gather_results — one all_gather of the adversarial clouds / best distances / labels at the end.
The same code runs on gloo (CPU tests, world_size 2).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous split of `total` samples: rank r gets [lo, hi). Sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _state_tensors(models):
    out = []
    for m in models:
        for _, t in sorted(m.state_dict().items()):
            out.append(t)
    return out


def broadcast_frozen_weights(models, src=0, group=None):
    """One collective for all floating-point parameters and buffers of `models` (one more for integer buffers,
    e.g. BatchNorm.num_batches_tracked, if any). In-place; invalidates folded-weight caches. Returns bytes sent."""
    tensors = _state_tensors(models)
    floats = [t for t in tensors if t.is_floating_point()]
    ints = [t for t in tensors if not t.is_floating_point()]
    nbytes = 0
    for group_t, dtype in ((floats, torch.float32), (ints, torch.int64)):
        if not group_t:
            continue
        flat = torch.cat([t.detach().reshape(-1).to(dtype) for t in group_t])
        dist.broadcast(flat, src=src, group=group)
        nbytes += flat.numel() * flat.element_size()
        off = 0
        with torch.no_grad():
            for t in group_t:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
                off += n
    for m in models:
        for sub in m.modules():
            if hasattr(sub, "_invalidate"):
                sub._invalidate()
    return nbytes


def gather_results(local_tensors, group=None):
    """all_gather a list of per-rank result tensors and concatenate along dim 0. Shards may differ in their first
    dimension (``shard_range`` hands out sizes that differ by one when total % world != 0): the per-rank lengths are
    exchanged first (one tiny all_gather), every rank pads to the longest shard for the equal-shape collective RCCL
    needs, and the padding is trimmed after it. Trailing dimensions and dtypes must agree across ranks.
    Returns the list of global tensors on every rank."""
    world = dist.get_world_size(group)
    if not local_tensors:
        return []
    dev = local_tensors[0].device
    mine = torch.tensor([t.shape[0] for t in local_tensors], dtype=torch.int64, device=dev)
    lens = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(lens, mine, group=group)
    lens = torch.stack(lens).cpu()                       # [world, n_tensors]
    out = []
    for j, t in enumerate(local_tensors):
        t = t.contiguous()
        longest = int(lens[:, j].max())
        if t.shape[0] < longest:
            t = torch.cat([t, t.new_zeros((longest - t.shape[0],) + tuple(t.shape[1:]))])
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out.append(torch.cat([p[:int(lens[r, j])] for r, p in enumerate(parts)], dim=0))
    return out
