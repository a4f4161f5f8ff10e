"""Walker sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The hot path is embarrassingly parallel over walkers: each rank evaluates a contiguous block of rows with its own
replica of the (tiny) model, and the only cross-rank step is the batch expectation of vqmc.py:196 -- one all-reduce
of three fp64 numbers [sum v, sum v^2, n] per step (24 B: latency-bound on xGMI, so a single in-place all-reduce
on the compute stream, no bucketing); a training step appends the flat gradient to that same buffer (~260 KB as fp64).
Walker coordinates never move between GPUs.
"""
import math


def _all_reduce_sum(t, group=None):
    """SUM all-reduce in place.  RCCL reduces device tensors directly; under gloo (CPU tests, or several ranks sharing one
    GPU in the test suite) a device tensor is reduced through a host copy."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_bounds(n_total, rank, world):
    """Rows [lo, hi) of rank `rank` when `n_total` walkers are cut into `world` contiguous blocks."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def all_reduce_moments(sums, group=None):
    """In-place SUM all-reduce of the fp64 triple produced by wf_block_sums (DeviceModel.block_sums).
    Works on any backend (nccl/RCCL on GPUs, gloo on CPU tensors); a no-op without an initialised group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce_sum(sums, group)
    return sums


def all_reduce_gradient_and_moments(grad, sums, group=None):
    """The training step's single collective: the flat gradient (already scaled by 1 / global batch) and the fp64
    triple travel in ONE packed fp64 buffer [gradient, sum, sum of squares, n].  -> (gradient float32, sums fp64)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return grad, sums
    packed = torch.cat([grad.double().reshape(-1), sums.double().reshape(-1)])
    _all_reduce_sum(packed, group)
    return packed[:-3].float(), packed[-3:]


def global_count(n_local, device, group=None):
    """Total walkers over the ranks of `group` (shards may be ragged)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return int(n_local)
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    _all_reduce_sum(cnt, group)
    return int(cnt.item())


def moments_to_stats(sums):
    """[sum, sum of squares, n] -> (mean, variance of the sample, standard error of the mean)."""
    s, q, n = (float(v) for v in sums)
    if n <= 0:
        return float("nan"), float("nan"), float("nan")
    mean = s / n
    var = max(q / n - mean * mean, 0.0)
    return mean, var, math.sqrt(var / n)


class ShardedDensity:
    """Evaluates log_pdf / psi on this rank's shard and reduces batch expectations across ranks."""

    def __init__(self, model, group=None):
        self.model = model
        self.group = group

    def local_rows(self, n_total):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return shard_bounds(n_total, dist.get_rank(self.group), dist.get_world_size(self.group))
        return 0, n_total

    def expectation(self, values):
        """values: this rank's fp32 device vector.  Returns the global (mean, variance, stderr)."""
        sums = self.model.block_sums(values)
        all_reduce_moments(sums, self.group)
        return moments_to_stats(sums.cpu().tolist())

    def mean_log_pdf(self, x_local):
        return self.expectation(self.model.log_pdf(x_local))
