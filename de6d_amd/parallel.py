"""Multi-GPU plumbing for the Det6D path: scenes are independent units, so inference shards them
across one process per GPU with NO collective on the data path.  What the
reference does with DistributedSampler(shuffle=False) + pickle files + two barriers
(core/pcdet/datasets/__init__.py:27-47,68-70; core/pcdet/utils/common_utils.py:212-233) is:
`scene_shard` + `gather_detections` (one all_gather_object at the very end).

`allreduce_gradients` is the optional training-side hook named by the north star: ONE flat
bucket, one RCCL all-reduce per step (9.4 MB of fp32 gradients for Det6D; over point-to-point
xGMI a single fused collective beats many small ones)."""
import torch
import torch.distributed as dist


def scene_shard(num_scenes, rank, world_size):
    """indices of the scenes rank `rank` processes: rank, rank + W, rank + 2W, ... padded by
    wrapping so every rank gets the same count (DistributedSampler(shuffle=False) semantics)."""
    per_rank = (num_scenes + world_size - 1) // world_size
    total = per_rank * world_size
    order = [i % num_scenes for i in range(total)]
    return order[rank:total:world_size]


def gather_detections(local_results, num_scenes, group=None):
    """local_results: list of per-scene results (any picklable object, e.g. dicts of numpy arrays) in
    the order of scene_shard(); returns the list for ALL scenes in scene order on every rank."""
    if not dist.is_available() or not dist.is_initialized():
        return list(local_results)[:num_scenes]
    world = dist.get_world_size(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, list(local_results), group=group)
    merged = []
    per_rank = len(gathered[0])
    for i in range(per_rank):          # interleave back: scene i*W + r came from rank r, slot i
        for r in range(world):
            merged.append(gathered[r][i])
    return merged[:num_scenes]


def allreduce_gradients(parameters, group=None, average=True):
    """one flat fp32 bucket -> one all-reduce -> scatter back (optional training step only)"""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads or not dist.is_initialized():
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel()
