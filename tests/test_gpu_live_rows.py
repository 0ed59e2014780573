"""Live rows (s2vt_teacher_forced_fwd_live / s2vt_bptt_bwd_live): a position behind a sample's first <eos> is masked
(cider_evaluation.py:145-172), its coefficient in the REINFORCE objective is zero, so the vocabulary projection, the softmax and
their two gradient products run on the unmasked (step, row) pairs only.  Logits of the live rows are bit-identical to the full
pass, the loss equal, gradients equal to the noise of the order-free reductions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=9)


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _case(oracle, B, rep, seed=3):
    d = oracle.Dims(label_dim=0, **DIMS)
    p = oracle.init_params(d, seed=seed)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    N = B * rep
    video = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    cap = rng.integers(2, d.n_words, (N, d.n_caption_lstm_step)).astype(np.int32)
    ln = rng.integers(0, d.n_caption_lstm_step - 1, N)
    ln[0] = d.n_caption_lstm_step - 2
    for n in range(N):
        cap[n, ln[n]:] = 0
    vid = np.tile(np.arange(B, dtype=np.int32) + 5, rep); sid = np.repeat(np.arange(rep, dtype=np.int32), B)
    return d, p, video, cap, vid, sid, N


@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_live_rows_match_the_full_pass(gpu, oracle, keep):
    import torch
    from s2vt_amd import hostglue
    d, p, video, cap, vid, sid, N = _case(oracle, B=4, rep=3)
    Tc = d.n_caption_lstm_step
    mask = hostglue.masks_from_ids(cap)
    steps = int(np.flatnonzero(mask.any(0))[-1]) + 1
    tm = mask[:, :steps].T.reshape(-1)
    live = np.flatnonzero(tm != 0).astype(np.int32)
    assert 0 < live.size < 0.8 * tm.size
    gd = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, Tc)
    dp_ = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp_)
    rng = np.random.default_rng(7)
    coef = (mask * rng.standard_normal(N)[:, None]).T.astype(np.float32).reshape(-1)[:steps * N]
    tgt = _dev(cap).t().contiguous().view(-1)[:steps * N]
    dl = _dev(live)

    def run(lv):
        logits, ws = gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, keep, 99, _dev(vid), _dev(sid), steps=steps, live=lv)
        raw = logits.clone()
        ix = slice(None) if lv is None else lv.long()
        nll, _ = gpu.softmax_nll_fwd_bwd(logits, tgt[ix].contiguous(), _dev(coef)[ix].contiguous(), 0.0)
        g = {k: torch.zeros_like(v) for k, v in dp_.items()}
        gpu.bptt_bwd(gd, params, gpu.make_params(g), _dev(video), N, logits, ws, keep, 99, _dev(vid), _dev(sid), steps=steps, live=lv)
        return raw, nll, g
    full, nll_f, g_f = run(None)
    part, nll_p, g_p = run(dl)
    assert part.shape[0] == live.size
    assert torch.equal(part, full[dl.long()])                      # the same chains, row by row
    assert torch.equal(nll_p, nll_f[dl.long()])
    for k in g_f:
        ref = g_f[k].cpu().numpy(); got = g_p[k].cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-12) + 1e-9, k


@pytest.mark.parametrize("fused", [False, True])
def test_reinforce_update_on_live_rows(gpu, oracle, fused):
    """model.reinforce_update: host mask -> live rows ("auto"), and the fused form (ids on the device, the caller passes the host
    mask it derived from the fetched ids, as train_rl does) against the dense pass: same loss, same variables after Adam."""
    import torch
    from s2vt_amd import hostglue, model as M
    d, p, video, cap, vid, sid, N = _case(oracle, B=4, rep=3)
    Tc = d.n_caption_lstm_step
    mask = hostglue.masks_from_ids(cap)
    rng = np.random.default_rng(5)
    r = rng.random(N).astype(np.float32) * 2; b = np.tile(rng.random(4).astype(np.float32) * 2, 3)
    outs = []
    for live in (False, True):
        mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, 4, 0, d.n_video_lstm_step, Tc, dropout_rate=0.9, seed=5)
        mdl.store.load(p)
        kw = dict(lr=1e-3, active_steps=None if not live else "auto", live_mask=None if not live else (mask if fused else "auto"))
        if fused and live:
            kw["active_steps"] = mdl.active_steps(mask)
        st = mdl.reinforce_update(video, cap, None if fused else mask, r, b, **kw)
        outs.append((float(st.loss), mdl.store.theta[:mdl.store.numel].clone(), mdl._ctx[9]))
    assert outs[0][2] is None and outs[1][2] is not None and outs[1][2].numel() == int((mask != 0).sum())
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * max(1.0, abs(outs[0][0]))
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-4


def test_live_rows_helper_thresholds():
    from s2vt_amd.model import Video_Caption_Generator as G
    import torch
    from s2vt_amd import ops
    mdl = G.__new__(G)
    mdl.device = torch.device("cuda")
    mdl.dims = ops.make_dims(128, 260, 32, 64, 5, 6)
    m = np.ones((4, 6), np.float32)
    assert mdl.live_rows(m, 6) is None                              # nothing masked: not worth a gather
    m[:, 3:] = 0
    assert mdl.live_rows(m, 6).tolist() == list(range(12))          # time-major: steps 0..2 of the four rows
    assert mdl.live_rows(m, 3) is None                              # inside the truncated unroll everything is live
    m[1, 1:] = 0
    assert mdl.live_rows(m, 3).tolist() == [0, 1, 2, 3, 4, 6, 7, 8, 10, 11]
    assert mdl.live_rows(torch.as_tensor(m).cuda(), 3) is None and mdl.live_rows(None, 3) is None
    m[1, 2] = 1                                                       # a hole: row 1 is masked at step 1 and live again at step 2
    assert mdl.live_rows(m, 3) is None                               # not a prefix per row: the dense pass


BIG = dict(dim_image=64, n_words=300, word_dim=32, lstm_dim=256, n_video_lstm_step=3, n_caption_lstm_step=9)


@pytest.mark.parametrize("B,rep,keep", [(64, 5, 0.9), (64, 6, 1.0), (66, 5, 1.0)])
def test_live_rows_stop_inside_the_recurrences(gpu, oracle, B, rep, keep):
    """257-384 rows (the register-weights recurrences, chain.hip / chain_bwd.hip): with a live list the rows are sorted by length
    on the device and a row tile behind every <eos> of its rows stops stepping, forward and backward.  Logits and loss terms of the
    live rows stay bit-identical to the dense pass (same chains), gradients equal to the noise of the order-free reductions -- and
    the launches that ran are the [live] variants."""
    import torch
    from s2vt_amd import hostglue, ops
    d = oracle.Dims(label_dim=0, **BIG)
    p = oracle.init_params(d, seed=11)
    rng = np.random.default_rng(12)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    N, Tc = B * rep, d.n_caption_lstm_step
    video = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    cap = rng.integers(2, d.n_words, (N, Tc)).astype(np.int32)
    ln = np.minimum(rng.poisson(2.5, N), Tc - 2)                 # mask length = ln + 1: short rows, a few long ones
    ln[7] = Tc - 2
    for n in range(N):
        cap[n, ln[n]:] = 0
    vid = np.tile(np.arange(B, dtype=np.int32) + 5, rep); sid = np.repeat(np.arange(rep, dtype=np.int32), B)
    mask = hostglue.masks_from_ids(cap)
    steps = int(np.flatnonzero(mask.any(0))[-1]) + 1
    tm = mask[:, :steps].T.reshape(-1)
    live = np.flatnonzero(tm != 0).astype(np.int32)
    assert live.size < 0.6 * tm.size
    gd = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, Tc)
    dp_ = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp_)
    coef = (mask * rng.standard_normal(N)[:, None]).T.astype(np.float32).reshape(-1)[:steps * N]
    tgt = _dev(cap).t().contiguous().view(-1)[:steps * N]
    dl = _dev(live)

    import contextlib

    def run(lv, hold_bwd=False):
        ops.prof_filter(-1, -1); ops.prof_enable(True)
        # every byte of the workspace starts as 0xFF (NaN as a float): a history slot of a stopped row that anything consumed shows
        ws = torch.full_like(ops.train_workspace(gd, B, N, torch.device("cuda")), 255)
        logits, ws = gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, keep, 99, _dev(vid), _dev(sid), steps=steps, live=lv, ws=ws)
        raw = logits.clone()
        ix = slice(None) if lv is None else lv.long()
        nll, _ = gpu.softmax_nll_fwd_bwd(logits, tgt[ix].contiguous(), _dev(coef)[ix].contiguous(), 0.0)
        g = {k: torch.zeros_like(v) for k, v in dp_.items()}
        with (ops.chain_hold() if hold_bwd else contextlib.nullcontext()):
            gpu.bptt_bwd(gd, params, gpu.make_params(g), _dev(video), N, logits, ws, keep, 99, _dev(vid), _dev(sid), steps=steps, live=lv)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        names = sorted(r["name"] for r in ops.prof_collect() if r["kernel_class"] in (5, 6))
        return raw, nll, g, names
    full, nll_f, g_f, names_f = run(None)
    part, nll_p, g_p, names_p = run(dl)
    # a collective in flight (s2vt_chain_hold) sends the backward recurrence to its per-step DENSE form over the workspace of a
    # forward pass that stopped rows: the unwritten slots hold the zeros the forward pass put there, the gradients are the same
    _, _, g_h, names_h = run(dl, hold_bwd=True)
    assert not any("[live]" in n for n in names_f)
    assert sum("chain4" in n and "[live]" in n for n in names_p) == 2, names_p        # forward and backward recurrence of LSTM2
    assert sum("[live]" in n for n in names_h) == 1 and not any(n.startswith("bchain") for n in names_h), names_h
    assert torch.equal(part, full[dl.long()])
    assert torch.equal(nll_p, nll_f[dl.long()])
    for k in g_f:
        ref = g_f[k].cpu().numpy()
        for got in (g_p[k].cpu().numpy(), g_h[k].cpu().numpy()):
            assert np.isfinite(got).all(), k
            assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-12) + 1e-9, k


def test_reinforce_update_at_320_rows_live_against_dense(gpu, oracle):
    """model.reinforce_update at K B = 320 rows (the shape whose recurrences stop rows, above) with the host mask ("auto" live rows,
    truncated unroll) against the dense full unroll: same loss, same finalised gradients, and the live variants of both
    register-weights recurrences ran."""
    import torch
    from s2vt_amd import hostglue, ops, model as M
    B, rep, Tc, Tv, H, E, V, D = 64, 5, 9, 3, 256, 32, 300, 64
    rng = np.random.default_rng(21)
    N = B * rep
    cap = rng.integers(1, V, (N, Tc)).astype(np.int32)
    ln = np.minimum(rng.poisson(2.5, N), Tc - 1)
    for n in range(N):
        cap[n, ln[n]:] = 0
    mask = hostglue.masks_from_ids(cap)
    video = np.abs(rng.standard_normal((B, Tv, D))).astype(np.float32)
    r = rng.random(N).astype(np.float32) * 2; b = np.tile(rng.random(B).astype(np.float32), rep)
    outs = []
    for live in (None, "auto"):
        mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, Tv, Tc, dropout_rate=0.9, seed=3)
        ops.prof_filter(-1, -1); ops.prof_enable(True)
        st = mdl.reinforce_update(video, cap, mask, r, b, lr=0.0, active_steps=None if live is None else "auto", live_mask=live)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        names = [x["name"] for x in ops.prof_collect() if x["kernel_class"] in (5, 6)]
        outs.append((float(st.loss), mdl.store.grad[:mdl.store.numel].clone(), names))
    assert not any("[live]" in n for n in outs[0][2])
    assert sorted(n for n in outs[1][2] if "[live]" in n) == ["bchain4(ng64,m320)[live]", "chain4(ng64,m320)[live]"], outs[1][2]
    g0, g1 = outs[0][1], outs[1][1]
    assert torch.isfinite(g1).all()
    assert float((g0 - g1).abs().max()) <= 5e-5 * float(g0.abs().max())
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * max(1.0, abs(outs[0][0]))


_INVALID_CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
sys.path.insert(0, os.path.join(os.environ["S2VT_ROOT"], "tests"))
import numpy as np, torch, pytest
import s2vt_amd
from s2vt_amd import hostglue, ops as gpu
from oracle import s2vt_oracle as oracle
from test_gpu_live_rows import _case, _dev
for defect in ("hole", "duplicate"):
    d, p, video, cap, vid, sid, N = _case(oracle, B=4, rep=3)
    Tc = d.n_caption_lstm_step
    mask = hostglue.masks_from_ids(cap)
    steps = int(np.flatnonzero(mask.any(0))[-1]) + 1
    on = mask[:, :steps] != 0
    n0 = int(np.argmax(on.sum(1)))                                    # the longest row: has >= 3 live steps
    assert on[n0].sum() >= 3
    if defect == "hole":
        on[n0, 1] = False                                             # steps {0, 2, ...}: position 2 sits behind a masked one
    live = np.flatnonzero(on.T.reshape(-1)).astype(np.int32)
    if defect == "duplicate":
        live = np.sort(np.concatenate([live, live[:1]])).astype(np.int32)
    gd = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, Tc)
    dp_ = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp_)
    assert not gpu.chain_fault()
    before = gpu.chain_timeouts()
    try:
        gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, 1.0, 99, _dev(vid), _dev(sid), steps=steps, live=_dev(live))
        torch.cuda.synchronize()
        assert gpu.chain_fault() and gpu.chain_timeouts() == before + 1, defect
        with pytest.raises(s2vt_amd._lib.S2VTChainTimeout):          # later passes are refused while the fault is pending
            gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, 1.0, 99, _dev(vid), _dev(sid), steps=steps)
        theta = torch.ones(64, device="cuda"); g = torch.ones(64, device="cuda"); m = torch.zeros(64, device="cuda"); v = torch.zeros(64, device="cuda")
        sq = torch.full((1,), 64.0, device="cuda")
        with pytest.raises(s2vt_amd._lib.S2VTChainTimeout):
            gpu.adam_tf(theta, g, m, v, sq, 5.0, 0.1, 1)
        assert float(theta.min()) == 1.0                              # no update went through
    finally:
        gpu.chain_ack(False)
    assert not gpu.chain_fault()
    # a valid list on the same shapes runs as before
    ok = np.flatnonzero((mask[:, :steps] != 0).T.reshape(-1)).astype(np.int32)
    logits, _ = gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, 1.0, 99, _dev(vid), _dev(sid), steps=steps, live=_dev(ok))
    torch.cuda.synchronize()
    assert not gpu.chain_fault() and logits.shape[0] == ok.size
print("child ok")
'''


def test_invalid_live_list_is_refused_loudly(gpu):
    """The *_live contract: the list is time-major, strictly ascending, and every row's live steps are a PREFIX of the decode steps
    (masks up to a first <eos>).  model.live_rows() checks that on the host; a C-API caller's device list is checked by the library's
    row_order_kernel -- a list with a hole (an unmasked position behind a masked one) or a duplicate would make LSTM2 stop a row too
    early and the gathered gradient products miss rows, silently.  It raises the sticky fault instead: queued updates skip on the device,
    every later entry point refuses, until s2vt_chain_ack.  (In a child process: the fault counter is process-wide and other tests assert
    that it reads 0.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _INVALID_CHILD], env=dict(os.environ, S2VT_ROOT=root), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "child ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
