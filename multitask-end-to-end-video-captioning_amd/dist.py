"""Data-parallel plumbing (NEW capability; the reference is single-GPU).

One process per GPU, full parameter replica.  Videos are the independent units: rank r takes
videos [r*B_loc, (r+1)*B_loc) and keeps its K samples local; the noise counters carry GLOBAL video
indices so 1 GPU x B and n GPUs x B/n draw the same tokens.  The only exchange per step is ONE
all-reduce (RCCL over xGMI when the backend is "nccl") of the flat, UNNORMALISED gradient bucket
with sum(mask) riding in its tail slot; every rank then applies 1/sum(mask), weight decay, the
global-norm clip and Adam identically, so no broadcast is needed.  These helpers are pure torch:
they are exercised on CPU with the gloo backend in tests/test_dp_gloo.py.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_videos: int, rank: int, world: int):
    """Contiguous shard of the global batch for this rank (requires n_videos % world == 0)."""
    assert n_videos % world == 0, "global batch must divide evenly over the ranks"
    per = n_videos // world
    return rank * per, (rank + 1) * per


def _all_reduce_sum(t: torch.Tensor, group=None):
    """SUM all-reduce in place.  Backend "nccl" (= RCCL on ROCm) reduces device memory directly; under
    "gloo" (CPU tests, and the single-GPU functional test of the N>1 path) a device tensor is staged
    through the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def allreduce_bucket(grad_flat: torch.Tensor, n_params: int, local_mask_sum, group=None):
    """In place: grad_flat[:n_params] <- sum over ranks, returns the GLOBAL sum(mask) as a 1-element
    view of the bucket's tail slot (so one collective carries both)."""
    grad_flat[n_params] = local_mask_sum
    if active(group):
        _all_reduce_sum(grad_flat, group)
    return grad_flat[n_params:n_params + 1]


def allreduce_small(t: torch.Tensor, group=None):
    if active(group):
        _all_reduce_sum(t, group)
    return t


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _forced() -> bool:
    import os
    return os.environ.get("S2VT_DP_FORCE", "0") == "1"


def active(group=None) -> bool:
    """True when the data-parallel exchange runs: more than one rank, or ONE rank with S2VT_DP_FORCE=1 -- which drives
    the whole collective path (RCCL load, async handles, cross-stream ordering) on a single GPU (tests/test_gpu_rccl.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _forced()


def allreduce_async(t: torch.Tensor, group=None):
    """Start a SUM all-reduce of `t` (a contiguous slice of the gradient bucket) and return a handle whose
    .wait() orders the CURRENT stream after it -- under "nccl" (RCCL) the collective runs on the communicator's
    own stream, beside the kernels launched meanwhile; under "gloo" it is done synchronously (host staging)."""
    if not active(group):
        return None
    if t.is_cuda and dist.get_backend(group) == "gloo":
        _all_reduce_sum(t, group)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
