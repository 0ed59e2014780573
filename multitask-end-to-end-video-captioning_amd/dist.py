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


# ---- optional timing of the exchange (bench.py's `allreduce_ms`): events on the CURRENT stream around each blocking
# all-reduce / each wait of an asynchronous one.  Under "nccl" the collective runs on the communicator's stream and the
# current stream waits for it, so the bracket is the time the step's stream spent on (or stalled behind) the exchange.
_timing = {"on": False, "pairs": []}


def timing_enable(on: bool):
    _timing["on"] = bool(on)
    _timing["pairs"].clear()


def timing_collect():
    """(total ms, brackets) since timing_enable(True); synchronise the device first."""
    ms = sum(a.elapsed_time(b) for a, b in _timing["pairs"])
    n = len(_timing["pairs"])
    _timing["pairs"].clear()
    return ms, n


class _Bracket:
    def __init__(self, t):
        self.on = _timing["on"] and t.is_cuda

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _timing["pairs"].append((self.e0, e1))


def _all_reduce_sum(t: torch.Tensor, group=None):
    """SUM all-reduce in place.  Backend "nccl" (= RCCL on ROCm) reduces device memory directly; under
    "gloo" (CPU tests, and the single-GPU functional test of the N>1 path) a device tensor is staged
    through the host."""
    with _Bracket(t):
        if t.is_cuda and dist.get_backend(group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def allreduce_bucket(grad_flat: torch.Tensor, n_params: int, local_mask_sum, group=None):
    """In place: grad_flat[:n_params] <- sum over ranks, returns the GLOBAL sum(mask) as a 1-element
    view of the bucket's tail slot (so one collective carries both).  local_mask_sum None: the slot already holds it."""
    if not active(group):
        # one process: nothing travels -- the local sum IS the global one, and copying it into the bucket's tail would be a 4-byte
        # runtime blit kernel per step (the only non-library launch the XE step's trace still showed, profiles/r04_xe_kernel_stats.md)
        if local_mask_sum is not None:
            return local_mask_sum.reshape(1) if isinstance(local_mask_sum, torch.Tensor) else grad_flat.new_full((1,), float(local_mask_sum))
        return grad_flat[n_params:n_params + 1]
    if local_mask_sum is not None:
        grad_flat[n_params] = local_mask_sum
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


def wait_all(handles, like: torch.Tensor):
    """Order the current stream after every pending asynchronous all-reduce (timed as one bracket)."""
    with _Bracket(like):
        for w in handles:
            if w is not None:
                w.wait()


def init_from_env(device=None):
    """One process per GPU under torch.distributed.run: read RANK / LOCAL_RANK / WORLD_SIZE, bind the GPU, create the
    process group (backend "nccl" = RCCL; S2VT_DIST_BACKEND=gloo only for functional tests that put several ranks on one
    GPU).  Returns (rank, world, device).  A no-op (0, 1, device) without WORLD_SIZE > 1."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if device is None:
        ndev = max(torch.cuda.device_count(), 1)
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
        if world > max(torch.cuda.device_count(), 1):
            # several ranks share a GPU (functional tests only): each rank's persistent recurrence needs ~every CU of the
            # chip, two of them in flight starve each other (csrc/chain.hip) -> per-step launches, same bits
            os.environ["S2VT_CHAIN"] = "0"
            os.environ["S2VT_BCHAIN"] = "0"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("S2VT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def replica_drift(theta: torch.Tensor, group=None) -> float:
    """max over ranks - min over ranks of two checksums of the flat variable buffer (sum and sum of squares, float64):
    data-parallel replicas apply identical updates to identical all-reduced gradients, so this must be exactly 0.0."""
    t64 = theta.double()
    cs = torch.stack([t64.sum(), (t64 * t64).sum()])
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0.0
    hi, lo = cs.clone(), cs.clone()
    if cs.is_cuda and dist.get_backend(group) == "gloo":
        hi, lo = hi.cpu(), lo.cpu()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return float((hi - lo).abs().max())
