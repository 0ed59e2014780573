"""Data-parallel equivalence ON the GPU kernels: two ranks x B/2 videos, all-reduced, end in the
same variables as one rank x B videos.  Both ranks share the one GPU of the test box, so the collective runs
over gloo (host staging) -- the product's nccl/RCCL path differs only in the all_reduce call."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6)
BG, K = 4, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    rng = np.random.default_rng(4)
    video = np.abs(rng.standard_normal((BG, 3, 24)) * 0.5).astype(np.float32)
    cap = rng.integers(0, 97, (K, BG, 6)).astype(np.int32); cap[..., -1] = 0
    r = rng.random((K, BG)).astype(np.float32) * 2; b = rng.random(BG).astype(np.float32) * 2
    return video, cap, r, b


def _run(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import s2vt_amd
    from s2vt_amd import hostglue, model as M
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    video, cap, r, b = _problem()
    per = BG // world
    lo, hi = rank * per, (rank + 1) * per
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, per, 0, 3, 6, dropout_rate=0.9, seed=9)
    mdl.world_size, mdl.rank = world, rank
    c = cap[:, lo:hi].reshape(K * per, -1)
    mask = hostglue.masks_from_ids(c)
    for step in range(2):
        mdl.reinforce_update(video[lo:hi], c, mask, r[:, lo:hi].reshape(-1), np.tile(b[lo:hi], K), lr=1e-2, clip_norm=5.0, video_base=lo)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out, mdl.store.theta.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import torch.multiprocessing as mp
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_run, args=(0, 1, 0, one)); p.start(); p.join(300); assert p.exitcode == 0
    port = _free_port()
    ps = [ctx.Process(target=_run, args=(rk, 2, port, two)) for rk in range(2)]
    [q.start() for q in ps]; [q.join(300) for q in ps]
    assert all(q.exitcode == 0 for q in ps)
    a, b = np.load(one), np.load(two)
    assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())


def _run_e2e(rank, world, port, out):
    """The CNN-in-the-loop step under data parallel: the CNN's flat gradient buffer is all-reduced too and both halves
    are clipped by the one global norm, so two ranks x B/2 must land where one rank x B does."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import s2vt_amd
    from s2vt_amd import e2e, hostglue, model as M
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(6)
    frames = rng.uniform(-1, 1, (BG, 3, 3, 15, 15)).astype(np.float32)
    cap = rng.integers(0, 97, (BG, 6)).astype(np.int32); cap[:, -2:] = 0
    per = BG // world
    lo, hi = rank * per, (rank + 1) * per
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, per, 0, 3, 6, dropout_rate=0.9, seed=9)
    mdl.world_size, mdl.rank = world, rank
    torch.manual_seed(3)
    cnn = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3, stride=2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                              torch.nn.Linear(6, 24), torch.nn.ReLU())
    tr = e2e.EndToEnd(mdl, cnn, feature_keep=0.9)          # feature dropout ON: masks keyed by the global video index
    mask = hostglue.masks_from_ids(cap[lo:hi])
    for step in range(2):
        tr.xe_step(torch.as_tensor(frames[lo:hi]), cap[lo:hi], mask, lr=1e-2, video_base=lo)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out, np.concatenate([mdl.store.theta.cpu().numpy(), tr.theta.cpu().numpy()]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_e2e_two_ranks_equal_one_rank(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import torch.multiprocessing as mp
    one, two = str(tmp_path / "e1.npy"), str(tmp_path / "e2.npy")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_run_e2e, args=(0, 1, 0, one)); p.start(); p.join(300); assert p.exitcode == 0
    port = _free_port()
    ps = [ctx.Process(target=_run_e2e, args=(rk, 2, port, two)) for rk in range(2)]
    [q.start() for q in ps]; [q.join(300) for q in ps]
    assert all(q.exitcode == 0 for q in ps)
    a, b = np.load(one), np.load(two)
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
