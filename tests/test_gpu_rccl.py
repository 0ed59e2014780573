"""The data-parallel exchange executed on RCCL on ONE MI355X.

The build box has a single GPU, so no scaling number can come from here; what CAN run is every line of the
collective path with a one-rank communicator: RCCL is loaded and initialised, the bucket travels through
ncclAllReduce, torch's async handles order the communicator's stream against the compute stream.

* world_size = 1 process group on backend "nccl" (= RCCL) with S2VT_DP_FORCE=1: reinforce_update / xe_update with the
  overlapped slices (backward phases 1 / 3 / 4, four async all-reduces) and with the single bucket all-reduce must
  leave the variables of a run without torch.distributed (to the run-to-run noise of the atomic reductions).
* s2vt_allreduce_grads (the C-host entry point) on a communicator made with ncclCommInitRank(nranks = 1) through
  ctypes, with the owning library's ncclAllReduce registered via s2vt_set_rccl_allreduce.
Both run in fresh child processes (a process group / communicator is process state).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD_DP = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import torch.distributed as dist
mode = sys.argv[1]
if mode != "plain":
    os.environ["S2VT_DP_FORCE"] = "1"
    os.environ["S2VT_DP_OVERLAP"] = "1" if mode == "overlap" else "0"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["S2VT_PORT"], world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
import s2vt_amd
from s2vt_amd import model as M, hostglue, dist as dp
assert dp.active() == (mode != "plain")
mdl = M.Video_Caption_Generator(128, 260, 32, 64, 4, 0, 5, 8, seed=5, dropout_rate=0.9)
assert mdl.dp_overlap == (mode == "overlap")
rng = np.random.default_rng(1)
video = torch.as_tensor(np.abs(rng.standard_normal((4, 5, 128)) * 0.5).astype(np.float32)).cuda()
K = 3
for it in range(3):
    s, _ = mdl.sample(video, K, True, seed=100 + it)
    mask = torch.as_tensor(hostglue.masks_from_ids(s.cpu().numpy())).cuda()
    r = (rng.random(K * 4) * 2).astype(np.float32); b = np.tile((rng.random(4) * 2).astype(np.float32), K)
    st = mdl.reinforce_update(video, s, mask, r, b, lr=1e-3, reuse_sampler_state=True)
cap = rng.integers(0, 260, (4, 8)).astype(np.int32); cap[:, -2:] = 0
st2 = mdl.xe_update(video, cap, hostglue.masks_from_ids(cap), lr=1e-3)
torch.cuda.synchronize()
np.save(sys.argv[2], np.concatenate([mdl.store.theta.cpu().numpy(), [float(st.loss), float(st2.loss)]]))
if mode != "plain":
    dist.destroy_process_group()
print("child ok", mode)
"""

CHILD_CABI = r"""
import ctypes as C, glob, os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import torch
import s2vt_amd
L = s2vt_amd.lib()
x = torch.arange(100000, dtype=torch.float32, device="cuda") * 0.5
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
# no RCCL entry point registered / visible yet: the call must refuse rather than load a second copy of the library
L.s2vt_set_rccl_allreduce(None)
cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")] + sorted(glob.glob("/opt/rocm/lib/librccl.so*"))
path = next(p for p in cands if os.path.exists(p))
R = C.CDLL(path, mode=os.RTLD_LOCAL)
class UID(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = UID()
R.ncclGetUniqueId.argtypes = [C.POINTER(UID)]
R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
assert R.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
assert R.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
assert L.s2vt_allreduce_grads(x.data_ptr(), x.numel(), None, st) == -1                 # NULL communicator
L.s2vt_set_rccl_allreduce(C.cast(R.ncclAllReduce, C.c_void_p))
ref = x.clone()
assert L.s2vt_allreduce_grads(x.data_ptr(), x.numel(), comm, st) == 0
torch.cuda.synchronize()
assert torch.equal(x, ref)                                                             # SUM over one rank
assert L.s2vt_allreduce_grads(x.data_ptr(), 0, comm, st) == 0
R.ncclCommDestroy.argtypes = [C.c_void_p]
R.ncclCommDestroy(comm)
print("child ok cabi", path)
"""


def _run(code, *args, port=None):
    env = dict(os.environ, S2VT_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("S2VT_DP_FORCE", None)
    if port:
        env["S2VT_PORT"] = str(port)
    r = subprocess.run([sys.executable, "-c", code, *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, f"rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def test_one_rank_rccl_process_group_matches_no_dist(gpu, tmp_path):
    import numpy as np
    outs = {}
    for i, mode in enumerate(("plain", "bucket", "overlap")):
        f = str(tmp_path / f"{mode}.npy")
        _run(CHILD_DP, mode, f, port=29611 + i)
        outs[mode] = np.load(f)
    # Same products in all three, but the order-free (atomic) weight-gradient reductions associate differently from run
    # to run, and Adam turns that noise into <= a few % of lr where g ~ 0.  A wrong exchange (a slice reduced twice or
    # never, a stale 1/sum(mask)) moves entries by ~lr per step: 4 steps x lr = 4e-3 against a 2e-4 bound.
    ref = outs["plain"]
    for mode in ("bucket", "overlap"):
        assert np.abs(outs[mode] - ref).max() <= 2e-4, mode


def test_c_host_allreduce_on_a_one_rank_communicator(gpu):
    _run(CHILD_CABI)


# ---- two ranks on two GPUs over RCCL (skipped on a one-GPU box; the driver's 8-GPU node and any >= 2-GPU box run it)
CHILD_DP2 = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import torch.distributed as dist
world, out = int(sys.argv[1]), sys.argv[2]
import s2vt_amd
from s2vt_amd import model as M, hostglue, dist as dp
rank, w, dev = dp.init_from_env()
assert w == world
if world > 1:
    assert dist.get_backend() == "nccl" and dp.active()
BG, K = 8, 3
rng = np.random.default_rng(1)
video = np.abs(rng.standard_normal((BG, 5, 128)) * 0.5).astype(np.float32)
cap = rng.integers(0, 260, (K, BG, 8)).astype(np.int32); cap[..., -1] = 0
r = (rng.random((K, BG)) * 2).astype(np.float32); b = (rng.random(BG) * 2).astype(np.float32)
gt = rng.integers(0, 260, (BG, 8)).astype(np.int32); gt[:, -2:] = 0
per = BG // world
lo, hi = rank * per, (rank + 1) * per
mdl = M.Video_Caption_Generator(128, 260, 32, 64, per, 0, 5, 8, seed=5, dropout_rate=0.9, device=dev)
mdl.world_size, mdl.rank = world, rank
c = cap[:, lo:hi].reshape(K * per, -1)
mask = hostglue.masks_from_ids(c)
for it in range(3):
    mdl.reinforce_update(video[lo:hi], c, mask, r[:, lo:hi].reshape(-1), np.tile(b[lo:hi], K), lr=1e-3, clip_norm=5.0, video_base=lo)
mdl.xe_update(video[lo:hi], gt[lo:hi], hostglue.masks_from_ids(gt[lo:hi]), lr=1e-3, video_base=lo)
torch.cuda.synchronize()
drift = dp.replica_drift(mdl.store.theta)
assert drift == 0.0, drift
if rank == 0:
    np.save(out, mdl.store.theta.cpu().numpy())
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("child ok", rank)
"""


def _run_ranks(code, world, out, port, overlap):
    env = dict(os.environ, S2VT_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               WORLD_SIZE=str(world), S2VT_DP_OVERLAP="1" if overlap else "0")
    env.pop("S2VT_DP_FORCE", None)
    env.pop("S2VT_DIST_BACKEND", None)
    ps = []
    for rk in range(world):
        e = dict(env, RANK=str(rk), LOCAL_RANK=str(rk))
        ps.append(subprocess.Popen([sys.executable, "-c", code, str(world), out], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in ps:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0 and "child ok" in so, f"rc={p.returncode}\n{so[-2000:]}\n{se[-4000:]}"


def test_two_rank_nccl_equals_one_rank(gpu, tmp_path):
    """2 ranks x B/2 on two GPUs over RCCL (one bucket all-reduce, and the overlapped slices) == 1 rank x B, and the two
    replicas hold bit-identical variables (checksum drift 0.0).  reinforcement_multisampling_tf_s2vt.py:643,650: the global
    sum(mask) and the clip come after the reduce."""
    import numpy as np
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    one = str(tmp_path / "one.npy")
    _run_ranks(CHILD_DP2, 1, one, 29631, False)
    ref = np.load(one)
    for i, overlap in enumerate((False, True)):
        two = str(tmp_path / f"two{i}.npy")
        _run_ranks(CHILD_DP2, 2, two, 29633 + i, overlap)
        got = np.load(two)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) + 2e-4, overlap
