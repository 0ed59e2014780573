"""The persistent LSTM recurrence (csrc/chain.hip: T steps in one launch, recurrent weights resident in LDS, state
exchanged through L2 behind a grid-wide hand-off; M <= 384 rows) against (1) T per-step launches of the fused cell kernel -- every state,
gate and dropped output BIT-identical -- and (2) the CPU oracle's BasicLSTMCell steps (tf_s2vt.py:113-153 restated).
Also under load on a second stream (hand-offs must not depend on timing) and repeated back to back (stale-state check:
every polled word and the fragment images are re-zeroed per call)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _case(M, E, H, T, csteps, seed):
    rng = np.random.default_rng(seed)
    W = rng.uniform(-.3, .3, (E + H, 4 * H)).astype(np.float32); b = rng.uniform(-.5, .5, 4 * H).astype(np.float32)
    h0 = rng.uniform(-1, 1, (M, H)).astype(np.float32); c0 = rng.standard_normal((M, H)).astype(np.float32)
    cinit = rng.standard_normal((max(csteps, 1), M, 4 * H)).astype(np.float32)
    vid = rng.integers(0, 1000, M).astype(np.int32); sid = rng.integers(0, 5, M).astype(np.int32)
    return W, b, h0, c0, cinit, vid, sid


SHAPES = [  # M, E, H, T, steps with a carried partial
    (4, 12, 20, 6, 3), (64, 32, 64, 9, 9), (33, 8, 64, 5, 0), (1, 4, 128, 4, 2),
    (64, 500, 1000, 25, 5),        # LSTM1 of the bench: B = 64, 5 frame steps with the hoisted input partial, 20 decode steps
    (32, 500, 1000, 25, 25),       # the multitask per-GPU batch, a partial at every step (LSTM2 form)
    (17, 500, 1000, 3, 1),
    (100, 8, 64, 4, 4), (130, 12, 128, 3, 0),   # two / three row tiles per wave
    # above 64 rows at H > 128, H % 8 == 0: the two-part form (8 units x half the row tiles per workgroup)
    (320, 500, 1000, 25, 25),                     # LSTM2 of build_loss at K * B = 320 rows: parts of 10 row tiles, 3 / 3 / 3 / 1 per wave
    (384, 500, 1000, 4, 2), (200, 500, 1000, 3, 3),   # 12 tiles per part (3 per wave); 13 row tiles: parts of 7 and 6
    (128, 500, 1000, 3, 3), (96, 16, 136, 3, 1),      # one row tile per wave; a part whose last waves are idle (3 tiles)
    # above 256 rows: the register-weights form (chain4: 16 units x a quarter of the row tiles per workgroup, A through LDS)
    (257, 16, 136, 3, 1), (300, 8, 1000, 3, 2), (336, 8, 1024, 3, 0), (384, 500, 1000, 25, 25), (321, 8, 264, 4, 4),
    (100, 8, 132, 3, 1),                              # H % 8 != 0: the one-part form at two row tiles per wave, 64 k groups
]


@pytest.mark.parametrize("M,E,H,T,csteps", SHAPES)
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_persistent_recurrence_equals_per_step_launches(gpu, oracle, M, E, H, T, csteps, keep):
    import torch
    W, b, h0, c0, cinit, vid, sid = _case(M, E, H, T, csteps, seed=M + H)
    args = dict(T=T, cinit=_dev(cinit) if csteps else None, cinit_steps=csteps, keep=keep, seed=77, video_id=_dev(vid),
                sample_id=_dev(sid), drop_code0=512, want_gates=True, want_out=True)
    ref = gpu.lstm_recurrence_fwd(_dev(W), E, _dev(b), _dev(h0), _dev(c0), persistent=0, **args)
    for rep in range(2):                                                   # back to back: nothing stale from the previous call
        got = gpu.lstm_recurrence_fwd(_dev(W), E, _dev(b), _dev(h0), _dev(c0), persistent=1, **args)
        assert gpu.chain_timeouts() == 0
        for name, r, g in zip(("C", "H", "gates", "out"), ref, got):
            assert torch.equal(r, g), (name, rep)
    # the oracle's own steps (small shapes: seconds on CPU)
    if H <= 128:
        c, h = c0, h0
        for t in range(T):
            mask = None if keep >= 1 else oracle.dropout_mask(77, vid, sid, 512 + t, keep, H)
            z = cinit[t].copy() if t < csteps else np.zeros((M, 4 * H), np.float32)       # the carried partial starts the chain
            oracle.gemm_chain(np.ascontiguousarray(h), np.ascontiguousarray(W[E:]), z)
            oracle.bias_add(z, b)
            rc, rh, rout, rg = oracle.lstm_pointwise(z, c, mask, keep, want_gates=True)
            assert np.array_equal(got[0][t + 1].cpu().numpy(), rc) and np.array_equal(got[1][t + 1].cpu().numpy(), rh), t
            assert np.array_equal(got[2][t].cpu().numpy(), rg) and np.array_equal(got[3][t].cpu().numpy(), rout), t
            c, h = rc, rh


@pytest.mark.parametrize("M_", [64, 320])
def test_persistent_recurrence_gates_over_the_partial_and_under_load(gpu, M_):
    """build_model's LSTM2 form: the activated gates overwrite the hoisted partial they continue from (G2 in train.hip);
    run while a second stream keeps the chip busy with unrelated kernels, several times -- same bits every time."""
    import torch
    M, E, H, T = M_, 500, 1000, 25
    W, b, h0, c0, cinit, vid, sid = _case(M, E, H, T, T, seed=5)
    dW, db, dh0, dc0 = _dev(W), _dev(b), _dev(h0), _dev(c0)
    ref = gpu.lstm_recurrence_fwd(dW, E, db, dh0, dc0, T, cinit=_dev(cinit), cinit_steps=T, keep=0.9, seed=3, video_id=_dev(vid),
                                  sample_id=_dev(sid), drop_code0=512, want_out=True, persistent=0, gates_in_cinit=True)
    side = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device="cuda")
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(3 + rep):
                x = torch.tanh(x @ x * 1e-3)                                  # uneven load beside the chain
        got = gpu.lstm_recurrence_fwd(dW, E, db, dh0, dc0, T, cinit=_dev(cinit), cinit_steps=T, keep=0.9, seed=3, video_id=_dev(vid),
                                      sample_id=_dev(sid), drop_code0=512, want_out=True, persistent=1, gates_in_cinit=True)
        torch.cuda.synchronize()
        assert gpu.chain_timeouts() == 0
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_), rep


def test_persistent_form_refuses_shapes_it_cannot_hold(gpu):
    import torch
    W = torch.zeros(8 + 6, 24, device="cuda"); b = torch.zeros(24, device="cuda")
    h0 = torch.zeros(3, 6, device="cuda")
    with pytest.raises(RuntimeError, match="bad argument"):
        gpu.lstm_recurrence_fwd(W, 8, b, h0, h0, 2, persistent=1)            # H = 6 is not a multiple of 4
    C, Hh, _, _ = gpu.lstm_recurrence_fwd(W, 8, b, h0, h0, 2, persistent=-1)  # auto: falls back to per-step launches
    assert C.shape == (3, 3, 6)
    W = torch.zeros(4 + 8, 32, device="cuda"); b = torch.zeros(32, device="cuda"); h0 = torch.zeros(385, 8, device="cuda")
    with pytest.raises(RuntimeError, match="bad argument"):
        gpu.lstm_recurrence_fwd(W, 4, b, h0, h0, 2, persistent=1)            # M = 385 rows: more than 6 row tiles per wave


# ---- a starved persistent recurrence must never corrupt a training run silently
CHILD_STARVED = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import s2vt_amd
from s2vt_amd import model as M, hostglue, ops
from s2vt_amd._lib import S2VTChainTimeout
from s2vt_amd.train_common import run_step
mode = sys.argv[1]
rng = np.random.default_rng(2)
B, K, Tv, Tc, D, V, E, H = 16, 2, 3, 6, 64, 300, 32, 1000
video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, Tv, Tc, seed=3, dropout_rate=0.9)
theta0 = mdl.store.theta.clone()
r = (rng.random(K * B) * 2).astype(np.float32); b = np.tile((rng.random(B) * 2).astype(np.float32), K)

def one_step(sync_ids):
    s, _ = mdl.sample(video, K, True, seed=50 + mdl.global_step)
    if sync_ids:
        s.cpu(); mdl.check_health()
    mask = ((torch.cumsum((s == 0).int(), 1) - (s == 0).int()) == 0).float()
    return mdl.reinforce_update(video, s, mask, r, b, lr=1e-2, reuse_sampler_state=True), s

if mode == "clean":                 # S2VT_CHAIN=0 in the environment: per-step launches, the reference result
    st, s = one_step(False); st2, s2 = one_step(False)
    torch.cuda.synchronize()
    assert ops.chain_timeouts() == 0 and not ops.chain_fault()
    np.savez(sys.argv[2], theta=mdl.store.theta.cpu().numpy(), ids=s2.cpu().numpy(), loss=float(st2.loss))
elif mode == "starved":             # S2VT_CHAIN_SPIN_LIMIT=0: the first grid-wide wait of every persistent launch gives up
    # (1) nothing synchronises inside the step (bench-like caller): the sampler's recurrence times out on the device, every
    # later entry point either refuses on the host or -- already queued -- skips on the device; the variables never move
    try:
        one_step(False); one_step(False)
        torch.cuda.synchronize()
        raised = False
    except S2VTChainTimeout:
        raised = True
    torch.cuda.synchronize()
    assert ops.chain_timeouts() > 0 and ops.chain_fault()
    assert torch.equal(mdl.store.theta, theta0), "a starved recurrence reached the variables"
    try:
        mdl.sample(video, K, True, seed=1)
        assert False, "the fault must be sticky"
    except S2VTChainTimeout:
        pass
    try:
        mdl.check_health(); assert False
    except S2VTChainTimeout:
        pass
    step, lost = mdl.recover()
    assert step == 0 and mdl.global_step == 0 and mdl.adam_t == 0 and not ops.chain_fault()
    # (2) the driver's loop (run_step): repeat from the intact variables on per-step launches -> the clean run's result
    n = ops.chain_timeouts()
    st, s = one_step(True); st2, s2 = one_step(True)
    torch.cuda.synchronize()
    assert ops.chain_timeouts() == n and not ops.chain_fault(), "the persistent form is off after recover()"
    np.savez(sys.argv[2], theta=mdl.store.theta.cpu().numpy(), ids=s2.cpu().numpy(), loss=float(st2.loss), raised=raised)
else:                               # "driver": run_step itself catches the fault, recovers and repeats the batch
    logs = []
    for i in range(2):
        st, loss = run_step(mdl, lambda: one_step(True)[0], logs.append)
    torch.cuda.synchronize()
    assert len(logs) == 1 and "timed out" in logs[0], logs
    assert mdl.global_step == 2 and not ops.chain_fault()
    s2, _ = mdl.sample(video, K, True, seed=50 + 1)       # (ids of step 2 are re-derivable only with the old weights: compare theta)
    np.savez(sys.argv[2], theta=mdl.store.theta.cpu().numpy(), loss=loss)
print("child ok", mode)
"""


def test_starved_recurrence_is_an_error_never_wrong_state(gpu, tmp_path):
    """A persistent recurrence that cannot get all its workgroups resident (another persistent kernel on the GPU) gives up
    its grid-wide wait.  Forced here with S2VT_CHAIN_SPIN_LIMIT=0.  Required: the variables are never updated from the
    garbage (Adam launches queued behind the fault skip on the device, later calls refuse on the host with
    S2VT_E_CHAIN_TIMEOUT), the fault is sticky until recover(), and after recover() the same steps on per-step launches
    give the result of a run that never used the persistent form."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, env in (("clean", {"S2VT_CHAIN": "0"}), ("starved", {"S2VT_CHAIN_SPIN_LIMIT": "0"}), ("driver", {"S2VT_CHAIN_SPIN_LIMIT": "0"})):
        out = str(tmp_path / f"{mode}.npz")
        e = dict(os.environ, S2VT_ROOT=root, **env)
        r = subprocess.run([sys.executable, "-c", CHILD_STARVED, mode, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "child ok" in r.stdout, f"{mode}: rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        res[mode] = dict(np.load(out))
    assert np.array_equal(res["clean"]["ids"], res["starved"]["ids"])                     # token ids: bit-exact, as always
    for mode in ("starved", "driver"):
        assert np.abs(res["clean"]["theta"] - res[mode]["theta"]).max() <= 5e-4, mode     # two Adam steps at lr 1e-2, atomic-order noise only
        assert abs(float(res["clean"]["loss"]) - float(res[mode]["loss"])) <= 1e-3


def test_persistent_launch_waits_for_the_previous_one_on_another_stream(gpu):
    """Two persistent grids of one process must never be in flight together (each needs ~every CU): a launch on a second
    stream is ordered behind the previous one by the library.  Both results bit-identical to per-step launches, no timeout."""
    import torch
    M_, E, H, T = 64, 500, 1000, 25
    W, b, h0, c0, cinit, vid, sid = _case(M_, E, H, T, 5, seed=5)
    args = dict(T=T, cinit=_dev(cinit), cinit_steps=5, keep=1.0, seed=7, video_id=_dev(vid), sample_id=_dev(sid), drop_code0=0,
                want_gates=True, want_out=False)
    ref = gpu.lstm_recurrence_fwd(_dev(W), E, _dev(b), _dev(h0), _dev(c0), persistent=0, **args)
    Wd, bd, hd, cd = _dev(W), _dev(b), _dev(h0), _dev(c0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):                 # (both streams also share ONE scratch buffer: overlapping launches would clobber it)
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(gpu.lstm_recurrence_fwd(Wd, E, bd, hd, cd, persistent=1, **args))
    torch.cuda.synchronize()
    assert gpu.chain_timeouts() == 0
    for got in outs[-2:]:
        for r, g in zip(ref[:3], got[:3]):
            assert torch.equal(r, g)


# ---- the persistent BACKWARD recurrence against per-step launches and against float64
BWD_SHAPES = [  # M, E, H, T, first step with an upstream gradient
    (4, 12, 20, 6, 2), (64, 32, 64, 9, 0), (33, 8, 64, 5, 3), (1, 4, 128, 4, 0),
    (64, 500, 1000, 25, 5),        # LSTM2 of the XE step / LSTM1 of every step at the bench dims
    (32, 500, 1000, 25, 0), (17, 500, 1000, 3, 1),
    (100, 8, 64, 4, 0), (128, 500, 1000, 6, 2),          # two row tiles per wave
    (200, 16, 136, 3, 0), (256, 500, 1000, 3, 1),        # four row tiles per wave (persistent = 1 only: auto stops at 128 rows)
    (64, 8, 1024, 3, 0),                                  # H = 1024: 64 unit groups x 4 gates = every CU
    # above 256 rows: the two-part form (32 units x half the row tiles per workgroup)
    (320, 500, 1000, 25, 5),                              # LSTM2 of build_loss at K * B = 320 rows: parts of 10 row tiles
    (384, 8, 1000, 3, 0), (260, 16, 136, 4, 1), (300, 8, 1024, 3, 1),
]


@pytest.mark.parametrize("M,E,H,T,t0", BWD_SHAPES)
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_persistent_backward_recurrence(gpu, M, E, H, T, t0, keep):
    """dZ of the one-launch backward recurrence == per-step {pointwise, split-K slabs} launches (to reduction order) and ==
    a float64 restatement of BasicLSTMCell's backward (tf.gradients through tf_s2vt.py:113-153) within 2e-5 of the largest
    entry per step; twice back to back (nothing stale), no timeout."""
    import torch
    W, b, h0, c0, cinit, vid, sid = _case(M, E, H, T, T, seed=M + H + 1)
    Wd = _dev(W)
    C, Hh, gates, _ = gpu.lstm_recurrence_fwd(Wd, E, _dev(b), _dev(h0), _dev(c0), T=T, cinit=_dev(cinit), cinit_steps=T, want_gates=True)
    rng = np.random.default_rng(M * 7 + H)
    dext = (rng.standard_normal((T - t0, M, H)) * 0.1).astype(np.float32)
    args = dict(dext=_dev(dext), dext_t0=t0, keep=keep, seed=99, video_id=_dev(vid), sample_id=_dev(sid), drop_code0=512)
    ref = gpu.lstm_recurrence_bwd(Wd, E, gates, C, persistent=0, **args)
    for rep in range(2):
        got = gpu.lstm_recurrence_bwd(Wd, E, gates, C, persistent=1, **args)
        assert gpu.chain_timeouts() == 0
        scale = ref.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
        assert float(((got - ref).abs() / scale).max()) <= 2e-5, rep
    if M <= 256 or H > 128:
        auto = gpu.lstm_recurrence_bwd(Wd, E, gates, C, persistent=-1, **args)
        assert torch.equal(auto, got)                       # auto takes the persistent form here (deterministic: no atomics)
    # float64 restatement
    g64 = gates.double().cpu().numpy(); C64 = C.double().cpu().numpy(); Whh = W[E:].astype(np.float64)
    dc = np.zeros((M, H)); dh_rec = np.zeros((M, H))
    for t in range(T - 1, -1, -1):
        dh = dh_rec.copy()
        if t >= t0:
            d = dext[t - t0].astype(np.float64)
            if keep < 1:
                mask = gpu_mask(gpu, 99, vid, sid, 512 + t, keep, H)
                d = d / np.float64(np.float32(keep)) * mask
            dh = dh + d
        si, tj, sf, so = (g64[t][:, q * H:(q + 1) * H] for q in range(4))
        tc = np.tanh(C64[t + 1])
        dcur = dh * so * (1 - tc * tc) + dc
        dz = np.concatenate([dcur * tj * si * (1 - si), dcur * si * (1 - tj * tj), dcur * C64[t] * sf * (1 - sf), dh * tc * so * (1 - so)], 1)
        dc = dcur * sf
        dh_rec = dz @ Whh.T
        err = np.abs(got[t].double().cpu().numpy() - dz).max() / max(np.abs(dz).max(), 1e-30)
        assert err <= 2e-5, (t, err)


def gpu_mask(gpu, seed, video, sample, code, keep, H):
    """DropoutWrapper keep masks [M, H] from the library's own stream (s2vt_dropout_bwd of ones = mask / keep; the stream
    itself is pinned to the oracle in test_gpu_fwd.py)."""
    import torch
    ones = torch.ones((len(video), H), dtype=torch.float32, device="cuda")
    return (gpu.dropout_bwd(ones, keep, seed, code, _dev(video), _dev(sample)) > 0).double().cpu().numpy()


CHILD_FORMS = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["S2VT_ROOT"], "tests"))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
from test_gpu_chain import _case, _dev
for (M, E, H, T, cs) in ((320, 500, 1000, 6, 6), (384, 8, 1000, 3, 1), (260, 16, 136, 3, 0)):
    W, b, h0, c0, cinit, vid, sid = _case(M, E, H, T, cs, seed=M)
    args = dict(T=T, cinit=_dev(cinit) if cs else None, cinit_steps=cs, keep=0.9, seed=3, video_id=_dev(vid), sample_id=_dev(sid), drop_code0=512,
                want_gates=True, want_out=True)
    ref = ops.lstm_recurrence_fwd(_dev(W), E, _dev(b), _dev(h0), _dev(c0), persistent=0, **args)
    got = ops.lstm_recurrence_fwd(_dev(W), E, _dev(b), _dev(h0), _dev(c0), persistent=1, **args)
    assert ops.chain_timeouts() == 0
    for r, g in zip(ref, got):
        assert torch.equal(r, g), (M, H)
    # an unaligned weight matrix (persistent = -1): per-step launches, same bits
    Wu = torch.empty(W.size + 1, dtype=torch.float32, device="cuda")[1:].view(W.shape); Wu.copy_(_dev(W))
    assert Wu.data_ptr() % 16 != 0
    auto = ops.lstm_recurrence_fwd(Wu, E, _dev(b), _dev(h0), _dev(c0), persistent=-1, **args)
    for r, g in zip(ref, auto):
        assert torch.equal(r, g), ("unaligned", M, H)
    # the backward recurrence in its two-part form (S2VT_BCHAIN4=0) against per-step launches
    C, Hh, gates, _ = ref
    dext = torch.as_tensor((np.random.default_rng(M).standard_normal((T, M, H)) * 0.1).astype(np.float32)).cuda()
    bargs = dict(dext=dext, dext_t0=0, keep=0.9, seed=5, video_id=_dev(vid), sample_id=_dev(sid), drop_code0=512)
    b0 = ops.lstm_recurrence_bwd(_dev(W), E, gates, C, persistent=0, **bargs)
    b1 = ops.lstm_recurrence_bwd(_dev(W), E, gates, C, persistent=1, **bargs)
    scale = b0.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
    assert float(((b1 - b0).abs() / scale).max()) <= 2e-5, ("bwd two-part", M, H)
    assert ops.chain_timeouts() == 0
print("child ok")
"""


def test_two_part_form_above_256_rows_and_unaligned_weights(gpu):
    """With the register-weights forms switched off (S2VT_CHAIN4=0, S2VT_BCHAIN4=0) the two-part forms serve 257-384 rows:
    forward still bit-identical to per-step launches, backward equal to reduction order.  And persistent = -1 with a weight matrix that is not 16-byte aligned falls back to per-step
    launches instead of failing."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD_FORMS], env=dict(os.environ, S2VT_ROOT=root, S2VT_CHAIN4="0", S2VT_BCHAIN4="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, f"rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"


def test_chain_hold_takes_the_per_step_form_and_the_same_bits(gpu):
    """ops.chain_hold(): while another stream's kernels share the GPU (an asynchronous all-reduce of a gradient slice,
    model.backward(overlap=True)), auto-selected recurrences must not start a persistent grid.  Inside the hold the launch profiler
    sees no persistent kernel (classes 5 / 6), outside it does; states, gates and dZ are bit-identical / equal either way, also with a
    second stream keeping the chip busy (standing in for RCCL's kernels)."""
    import torch
    ops = gpu
    torch.manual_seed(3)
    M_, H, T = 64, 1000, 6
    W = (torch.rand(H + 8, 4 * H, device="cuda") * 2 - 1) * 0.1
    b = torch.zeros(4 * H, device="cuda")
    h0 = torch.zeros(M_, H, device="cuda"); c0 = torch.zeros(M_, H, device="cuda")
    cin = (torch.rand(T, M_, 4 * H, device="cuda") * 2 - 1)

    def run():
        ops.prof_filter(-1, -1); ops.prof_enable(True)
        Ch, Hh, gates, _ = ops.lstm_recurrence_fwd(W, 8, b, h0, c0, T, cinit=cin, cinit_steps=T)
        dZ = ops.lstm_recurrence_bwd(W, 8, gates, Ch, dext=torch.ones(T, M_, H, device="cuda"))
        torch.cuda.synchronize()
        rows = ops.prof_collect(); ops.prof_enable(False)
        return Ch, Hh, gates, dZ, {r["kernel_class"] for r in rows}
    free = run()
    hog = torch.cuda.Stream()
    big = torch.rand(4096, 4096, device="cuda")
    with ops.chain_hold():
        with torch.cuda.stream(hog):
            for _ in range(20):
                big = (big @ big).clamp_(-1, 1)            # a few ms of every CU busy on another stream
        held = run()
    torch.cuda.synchronize()
    again = run()
    assert 5 in free[4] and 6 in free[4] and 5 in again[4], (free[4], again[4])      # the persistent forms, before and after
    assert 5 not in held[4] and 6 not in held[4], held[4]                            # ... and not while held
    for a, c in zip(free[:3], held[:3]):
        assert torch.equal(a, c)
    assert float((free[3] - held[3]).abs().max()) <= 2e-5 * float(free[3].abs().max())
    assert ops.chain_timeouts() == 0
