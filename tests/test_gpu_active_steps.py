"""Truncated unroll (s2vt_teacher_forced_fwd_steps / s2vt_bptt_bwd_steps): behind the longest caption of a batch every
position is masked (tf_s2vt.py:371-401 pads to Tc; cider_evaluation.py:145-172 masks a sample behind its first <eos>), so
those steps add exact zeros to the loss and to every gradient.  The truncated pass must reproduce the full one: logits
bit-identical on the steps it runs, gradients equal to the noise of the order-free reductions, the same update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = [
    dict(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6),
    dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=9),
]


def _dev(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _setup(oracle, dims, B, rep, longest, seed=3):
    d = oracle.Dims(label_dim=0, **dims)
    p = oracle.init_params(d, seed=seed)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    N = B * rep
    video = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    cap = rng.integers(2, d.n_words, (N, d.n_caption_lstm_step)).astype(np.int32)
    ln = rng.integers(0, longest, N)                      # words before <eos>; the longest row has `longest - 1`
    ln[0] = longest - 1
    for n in range(N):
        cap[n, ln[n]:] = 0
    vid = np.tile(np.arange(B, dtype=np.int32) + 5, rep); sid = np.repeat(np.arange(rep, dtype=np.int32), B)
    return d, p, video, cap, vid, sid, N


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_truncated_pass_is_the_leading_part_of_the_full_one(gpu, oracle, dims, keep):
    import torch
    from s2vt_amd import hostglue
    Tc = dims["n_caption_lstm_step"]
    steps = Tc - 2
    d, p, video, cap, vid, sid, N = _setup(oracle, dims, B=3, rep=2, longest=steps)
    mask = hostglue.masks_from_ids(cap)
    assert mask[:, steps:].sum() == 0 and mask[:, steps - 1].sum() > 0
    gd = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, Tc)
    dp_ = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp_)
    rng = np.random.default_rng(7)
    coef = (mask * rng.standard_normal(N)[:, None]).T.astype(np.float32).reshape(-1)          # time-major, zero where masked
    tgt = _dev(cap).t().contiguous().view(-1)

    def run(s):
        R = (Tc if s is None else s) * N
        logits, ws = gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, keep, 99, _dev(vid), _dev(sid), steps=s)
        raw = logits.clone()
        nll, _ = gpu.softmax_nll_fwd_bwd(logits, tgt[:R], _dev(coef)[:R], 0.0)
        g = {k: torch.zeros_like(v) for k, v in dp_.items()}
        gpu.bptt_bwd(gd, params, gpu.make_params(g), _dev(video), N, logits, ws, keep, 99, _dev(vid), _dev(sid), steps=s)
        return raw, nll, g
    full, nll_f, g_f = run(None)
    part, nll_p, g_p = run(steps)
    assert part.shape[0] == steps * N
    assert torch.equal(part, full[:steps * N])                       # same ascending-k chains on the steps that run
    assert torch.equal(nll_p, nll_f[:steps * N])
    for k in g_f:
        ref = g_f[k].cpu().numpy(); got = g_p[k].cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-12) + 1e-9, k


def test_steps_argument_is_checked(gpu, oracle):
    import s2vt_amd
    dims = DIMS[0]
    d, p, video, cap, vid, sid, N = _setup(oracle, dims, B=3, rep=1, longest=3)
    gd = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    params = gpu.make_params({k: _dev(v) for k, v in p.items()})
    for bad in (0, d.n_caption_lstm_step + 1):
        import torch
        logits = torch.empty((bad * N, d.n_words), dtype=torch.float32, device="cuda")
        with pytest.raises(s2vt_amd.S2VTLibraryError):
            gpu.teacher_forced_fwd(gd, params, _dev(video), _dev(cap), N, 1.0, 0, None, None, logits=logits, steps=bad)


@pytest.mark.parametrize("mode", ["xe_q1", "xe_plain", "pg", "pg_fused_ids", "mixed"])
def test_updates_with_and_without_the_padding_steps(gpu, oracle, mode):
    """xe_update / reinforce_update / mixed_update: active_steps="auto" (host-resident masks) against the full unroll --
    same loss, same variables after one Adam step with dropout on."""
    import torch
    from s2vt_amd import hostglue, model as M
    dims = DIMS[1]
    Tc = dims["n_caption_lstm_step"]
    B, rep = 4, (1 if mode.startswith("xe") else 3)
    d, p, video, cap, vid, sid, N = _setup(oracle, dims, B=B, rep=rep, longest=Tc - 3)
    mask = hostglue.masks_from_ids(cap)
    rng = np.random.default_rng(5)
    r = rng.random(N).astype(np.float32) * 2; b = np.tile(rng.random(B).astype(np.float32) * 2, rep)
    gcap = cap[:B].copy(); gmask = mask[:B].copy()
    outs = []
    for active in (None, "auto"):
        mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, B, 0, d.n_video_lstm_step, Tc,
                                        dropout_rate=0.9, seed=5)
        mdl.store.load(p)
        if mode == "xe_q1":
            st = mdl.xe_update(video, cap, mask, lr=1e-3, q1=True, active_steps=active)
        elif mode == "xe_plain":                           # (without Q1 the second run also packs the masked positions out: live rows)
            st = mdl.xe_update(video, cap, mask, lr=1e-3, q1=False, active_steps=active, live_mask=None if active is None else "auto")
            assert (mdl._ctx[9] is not None) == (active is not None)
        elif mode == "pg":
            st = mdl.reinforce_update(video, cap, mask, r, b, lr=1e-3, active_steps=active, live_mask=None)
        elif mode == "pg_fused_ids":                     # mask derived on the device: the caller passes the count (train_rl does)
            st = mdl.reinforce_update(video, cap, None, r, b, lr=1e-3, active_steps=None if active is None else M.Video_Caption_Generator.active_steps(mask))
        else:
            st = mdl.mixed_update(video, cap if active else _dev(cap), mask if active else _dev(mask), r, b, gcap, gmask, lr=1e-3, active_steps=active)
            assert (mdl._ctx[9] is not None) == (active is not None)      # host masks: live rows; device masks: the dense pass
        ctx_steps = mdl._ctx[8]
        outs.append((float(st.loss), mdl.store.theta[:mdl.store.numel].clone(), ctx_steps))
    assert outs[0][2] == Tc and outs[1][2] == Tc - 3            # the second run really skipped the padding
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * max(1.0, abs(outs[0][0]))
    # Adam at lr 1e-3 moves every variable by ~lr whatever its gradient's size: compare on that scale
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-4


def test_active_steps_helper():
    import torch
    from s2vt_amd.model import Video_Caption_Generator as G
    m = np.zeros((3, 6), np.float32)
    assert G.active_steps(m) == 1                                 # nothing live: one step is the minimum unroll
    m[1, :4] = 1
    assert G.active_steps(m) == 4
    m[2, 5] = 1                                                    # any live position counts, prefix-shaped or not
    assert G.active_steps(m) == 6
    assert G.active_steps(torch.as_tensor(m)) == 6
    assert G.active_steps(torch.as_tensor(m).cuda()) is None       # a device mask is not synchronised for this
    assert G.active_steps(None) is None
