"""Data-parallel path on CPU: world_size-2 gloo processes.  The per-rank compute is the torch
oracle (the HIP kernels need a GPU); what is under test is the product's DP plumbing
(s2vt_amd.dist): unnormalised gradient bucket + sum(mask) in one all-reduce, then identical
normalisation on every rank == the single-process gradient of the global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    sys.path.insert(0, ROOT)
    from oracle import s2vt_oracle as orc
    d = orc.Dims(dim_image=10, n_words=29, word_dim=6, lstm_dim=8, n_video_lstm_step=3, n_caption_lstm_step=4, label_dim=0)
    p = orc.init_params(d, 5)
    rng = np.random.default_rng(8)
    Bg, K = 4, 2
    video = np.abs(rng.standard_normal((Bg, 3, 10))).astype(np.float32)
    cap = rng.integers(0, 29, (K, Bg, 4)).astype(np.int32)          # [k, j, t]
    mask = (rng.random((K, Bg, 4)) < .7).astype(np.float32); mask[..., 0] = 1
    r = rng.random((K, Bg)); b = rng.random(Bg)
    return d, p, video, cap, mask, r, b, Bg, K


def _unnormalised_grads(p, video, cap, mask, r, b):
    """sum_{n,t} -lp*mask*(r-b) (no 1/sum(mask)) and its gradient, float64 torch oracle."""
    from oracle import s2vt_torch as T
    pt = T.to_torch(p, torch.float64, True)
    K, B = cap.shape[0], cap.shape[1]
    capf = cap.reshape(K * B, -1); maskf = mask.reshape(K * B, -1)
    logits = T.teacher_forced(pt, torch.as_tensor(np.tile(video, (K, 1, 1))).double(), capf)
    loss = T.pg_loss(logits, capf, maskf, r.reshape(-1), np.tile(b, K)) * float(maskf.sum())
    loss.backward()
    names = sorted(pt)
    flat = torch.cat([pt[n].grad.reshape(-1) for n in names])
    return flat, float(maskf.sum())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import s2vt_amd
    from s2vt_amd import dist as dp
    d, p, video, cap, mask, r, b, Bg, K = _problem()
    lo, hi = dp.shard_range(Bg, rank, world)
    g, msum = _unnormalised_grads(p, video[lo:hi], cap[:, lo:hi], mask[:, lo:hi], r[:, lo:hi], b[lo:hi])
    bucket = torch.zeros(g.numel() + 64, dtype=torch.float64)
    bucket[:g.numel()] = g
    gsum = dp.allreduce_bucket(bucket, g.numel(), msum)
    col = dp.allreduce_small(torch.as_tensor(mask[:, lo:hi].sum((0, 1))))
    out[rank] = ((bucket[:g.numel()] / gsum).numpy(), float(gsum), col.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_equals_single_process_gradient():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d, p, video, cap, mask, r, b, Bg, K = _problem()
    g, msum = _unnormalised_grads(p, video, cap, mask, r, b)
    ref = (g / msum).numpy()
    for rank in range(world):
        got, gsum, col = out[rank]
        assert abs(gsum - msum) < 1e-9
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)       # every rank holds the global-batch gradient
        assert np.allclose(col, mask.sum((0, 1)))


def test_shard_range_and_single_process_bucket():
    sys.path.insert(0, ROOT)
    import s2vt_amd
    from s2vt_amd import dist as dp
    assert [dp.shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(AssertionError):
        dp.shard_range(10, 0, 4)
    bucket = torch.arange(10, dtype=torch.float32)
    s = dp.allreduce_bucket(bucket, 6, 3.5)                       # no process group: identity + tail slot
    assert float(s) == 3.5 and bucket[:6].tolist() == [0, 1, 2, 3, 4, 5]
