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


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks ITSELF (the driver's N>1
    command is torch.distributed.run, but a bare invocation may not silently run one rank) and print a line that proves
    it: n_gpus 2, the backend, replica drift 0.0, an all-reduce time.  Two ranks share this box's one GPU, so the
    functional backend is gloo here (rccl_ranks reports 0 for it: not an RCCL run)."""
    import json
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    nccl = torch.cuda.device_count() >= 2
    env["S2VT_DIST_BACKEND"] = "nccl" if nccl else "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["parallelism"] == "dp2" and cfg["global_batch"] == 128
    assert cfg["dist_backend"] == ("nccl" if nccl else "gloo") and cfg["rccl_ranks"] == (2 if nccl else 0)
    assert cfg["replica_max_abs_diff"] == 0.0
    assert cfg["allreduce_ms"] > 0.0 and cfg["persistent_recurrence_timeouts"] == 0
    assert "cpu_baseline" not in out                       # rank 0 at N = 1 only
