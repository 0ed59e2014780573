"""GPU parity, update half of the step: teacher-forced forward (bit-exact incl. Philox dropout),
softmax-NLL, BPTT gradients vs float64 autograd of the torch restatement, clip + TF-Adam."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(B=3, rep=2, dims=dict(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6)),
    dict(B=4, rep=3, dims=dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=8)),
]


def _dev(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _setup(oracle, case, seed=3):
    d = oracle.Dims(label_dim=0, **case["dims"])
    p = oracle.init_params(d, seed=seed)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    B, rep = case["B"], case["rep"]
    N = B * rep
    video = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    cap = rng.integers(0, d.n_words, (N, d.n_caption_lstm_step)).astype(np.int32)
    cap[:, -2:] = 0
    cap[1, 2] = 0
    vid = np.tile(np.arange(B, dtype=np.int32) + 5, rep); sid = np.repeat(np.arange(rep, dtype=np.int32), B)
    return d, p, video, cap, vid, sid, N


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_teacher_forced_logits_bit_exact(gpu, oracle, case, keep):
    d, p, video, cap, vid, sid, N = _setup(oracle, case)
    drop = None if keep >= 1 else oracle.dropout_masks(99, vid, sid, keep, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    ref = oracle.teacher_forced(p, d, np.tile(video, (case["rep"], 1, 1)), cap, drop, keep)       # [N,Tc,V]
    dims = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    dp = {k: _dev(v) for k, v in p.items()}
    logits, _ = gpu.teacher_forced_fwd(dims, gpu.make_params(dp), _dev(video), _dev(cap), N, keep, 99, _dev(vid), _dev(sid))
    got = logits.view(d.n_caption_lstm_step, N, d.n_words).permute(1, 0, 2).cpu().numpy()
    assert np.array_equal(got, ref)


def test_softmax_nll_rows(gpu, oracle):
    import torch
    rng = np.random.default_rng(0)
    for R, V, s in [(7, 23, 0.0), (33, 260, 0.05), (5, 12000, 0.05), (9, 9972, 0.0)]:
        logits = (3 * rng.standard_normal((R, V))).astype(np.float32)
        tgt = rng.integers(0, V, R).astype(np.int32); coef = rng.standard_normal(R).astype(np.float32)
        nll_ref, lp_ref, _ = oracle.row_losses(logits, tgt, s)
        dl = _dev(logits)
        nll, lp = gpu.softmax_nll_fwd_bwd(dl, _dev(tgt), _dev(coef), s)
        assert np.allclose(nll.cpu().numpy(), nll_ref, rtol=1e-5, atol=1e-5)      # reduction order differs: tolerance
        assert np.allclose(lp.cpu().numpy(), lp_ref, rtol=1e-5, atol=1e-5)
        lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
        lpt = torch.log_softmax(lt, -1)
        q = torch.full_like(lpt, s / V); q[torch.arange(R), torch.as_tensor(tgt).long()] += 1 - s
        (-(q * lpt).sum(-1) * torch.tensor(coef, dtype=torch.float64)).sum().backward()
        assert np.allclose(dl.cpu().numpy(), lt.grad.numpy(), rtol=1e-4, atol=1e-6)
        assert np.abs(dl.sum(1).cpu().numpy()).max() < 1e-4                      # rows of (softmax - q) sum to 0


def _torch_grads(oracle, p, d, video_t, cap, drop, keep, loss_fn):
    import torch
    from oracle import s2vt_torch as T
    pt = T.to_torch(p, torch.float64, True)
    logits = T.teacher_forced(pt, torch.as_tensor(video_t).double(), cap, drop, keep)
    loss = loss_fn(pt, logits)
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in pt.items()}


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["pg", "xe_q1", "xe_plain"])
def test_gradients_vs_float64_autograd(gpu, oracle, case, mode):
    """d(objective)/d(every variable): HIP BPTT vs float64 autograd over the restated graph, with the
    SAME Philox dropout masks.  Tolerance 2e-4 relative to each tensor's largest gradient entry."""
    import torch
    import s2vt_amd
    from s2vt_amd import model as M
    from oracle import s2vt_torch as T
    d, p, video, cap, vid, sid, N = _setup(oracle, case)
    rep, keep = case["rep"], 0.9
    rng = np.random.default_rng(5)
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    r = rng.random(N).astype(np.float32) * 2; b = np.tile(rng.random(case["B"]).astype(np.float32) * 2, rep)
    mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, case["B"], 0, d.n_video_lstm_step,
                                    d.n_caption_lstm_step, dropout_rate=keep)
    mdl.store.load(p)
    if mode != "pg":
        rep = 1
        video_rows = np.abs(rng.standard_normal((N, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
        vid = np.arange(N, dtype=np.int32) + 5; sid = np.zeros(N, np.int32)
    else:
        video_rows = np.tile(video, (rep, 1, 1))
    dseed = mdl.dropout_seed + 104729 * mdl.global_step
    drop = oracle.dropout_masks(dseed, vid, sid, keep, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    if mode == "pg":
        fn = lambda pt, lg: T.pg_loss(lg, cap, mask, r, b)
    else:
        fn = lambda pt, lg: T.xe_loss(pt, lg, cap, mask, q1=(mode == "xe_q1"))
    ref_loss, ref_g = _torch_grads(oracle, p, d, video_rows, cap, drop, keep, fn)

    # run the product's update with lr = 0 so the variables stay put, then read the finalized gradients
    if mode == "pg":
        st = mdl.reinforce_update(video, cap, mask, r, b, lr=0.0, clip_norm=5.0, video_base=5)
        loss = float(st.loss)
    else:
        st = mdl.xe_update(video_rows, cap, mask, lr=0.0, clip_norm=10.0, q1=(mode == "xe_q1"), video_base=5)
        wd = sum(0.5 * float((mdl.store.p[n].double() ** 2).sum()) for n in mdl.store.names if n not in M.UNDECAYED)
        loss = float(st.loss) + mdl.decay_value * wd
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    gn = 0.0
    for n in mdl.store.names:
        g = mdl.store.g[n].cpu().numpy().astype(np.float64)
        scale = np.abs(ref_g[n]).max() + 1e-12
        assert np.abs(g - ref_g[n]).max() <= 2e-4 * scale + 1e-9, (n, np.abs(g - ref_g[n]).max(), scale)
        gn += (ref_g[n] ** 2).sum()
    assert abs(float(st.grad_sumsq) - gn) <= 1e-3 * gn            # the global norm the clip will see


def test_clip_and_adam_tf_three_steps(gpu, oracle):
    """tf.clip_by_global_norm + tf.train.AdamOptimizer (epsilon outside the bias correction)."""
    import torch
    from oracle import s2vt_torch as T
    rng = np.random.default_rng(1)
    n = 5000
    theta = rng.standard_normal(n).astype(np.float32)
    pt = {"w": torch.tensor(theta, dtype=torch.float64)}
    mt = {"w": torch.zeros(n, dtype=torch.float64)}; vt = {"w": torch.zeros(n, dtype=torch.float64)}
    th = _dev(theta); m = torch.zeros_like(th); v = torch.zeros_like(th)
    for step in range(1, 4):
        g = (rng.standard_normal(n) * (10.0 if step == 2 else 0.01)).astype(np.float32)   # step 2 gets clipped
        gt, _ = T.clip_by_global_norm({"w": torch.tensor(g, dtype=torch.float64) * 0.5 + 1e-3 * pt["w"]}, 5.0)
        pt, mt, vt = T.adam_tf(pt, gt, mt, vt, step, 1e-3)
        gd = _dev(g); sumsq = torch.zeros(1, device="cuda"); gscale = torch.full((1,), 0.5, device="cuda")
        gpu.grad_finalize(gd, th, gscale, 1e-3, sumsq)
        gpu.adam_tf(th, gd, m, v, sumsq, 5.0, 1e-3, step)
        assert np.allclose(th.cpu().numpy(), pt["w"].numpy(), rtol=2e-5, atol=2e-6)
    assert np.allclose(m.cpu().numpy(), mt["w"].numpy(), rtol=1e-4, atol=1e-7)


def test_sampler_state_reuse_is_equivalent(gpu, oracle):
    """reinforce_update(reuse_sampler_state=True) takes LSTM1's trajectory from the sampler pass of the step: the
    loss is bit-identical to recomputing it, the finalized gradients agree to the run-to-run noise of the
    order-free (atomic) gradient reductions."""
    import torch
    from s2vt_amd import hostglue, model as M
    d = oracle.Dims(label_dim=0, **CASES[1]["dims"])
    rng = np.random.default_rng(11)
    B, K = 4, 3
    video = torch.as_tensor(np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)).cuda()
    r = rng.random(K * B).astype(np.float32) * 2; b = np.tile(rng.random(B).astype(np.float32) * 2, K)
    outs = []
    for reuse in (False, True):
        mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, B, 0, d.n_video_lstm_step,
                                        d.n_caption_lstm_step, dropout_rate=0.9, seed=5)
        s, _ = mdl.sample(video, K, True, seed=77)
        mask = torch.as_tensor(hostglue.masks_from_ids(s.cpu().numpy())).cuda()
        st = mdl.reinforce_update(video, s, mask, r, b, lr=0.0, reuse_sampler_state=reuse)
        outs.append((float(st.loss), mdl.store.grad[:mdl.store.numel].clone()))
    assert outs[0][0] == outs[1][0]
    scale = float(outs[0][1].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-5 * scale
    # the mixed multitask objective (sampled + ground-truth rows in one pass): LSTM1's trajectory is per VIDEO, the same for both row blocks
    gcap = rng.integers(2, d.n_words, (B, d.n_caption_lstm_step)).astype(np.int32); gcap[:, -2:] = 0
    gmask = hostglue.masks_from_ids(gcap)
    outs = []
    for reuse in (False, True):
        mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, B, 0, d.n_video_lstm_step,
                                        d.n_caption_lstm_step, dropout_rate=0.9, seed=5, multisample=K)
        s, _ = mdl.sample(video, K, True, seed=78)
        mask = hostglue.masks_from_ids(s.cpu().numpy())
        st = mdl.mixed_update(video, s, mask, r, b, gcap, gmask, lr=0.0, reuse_sampler_state=reuse)
        outs.append((float(st.loss), mdl.store.grad[:mdl.store.numel].clone()))
    assert outs[0][0] == outs[1][0]
    scale = float(outs[0][1].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-5 * scale
    with pytest.raises(RuntimeError):                                  # no sampler call on THIS tensor with the current weights: refused
        mdl.mixed_update(video.clone(), s, mask, r, b, gcap, gmask, lr=0.0, reuse_sampler_state=True)


def test_mixed_pg_xe_objective_vs_float64_autograd(gpu, oracle):
    """reinforce_multitask_e2e_attribute_s2vt.py:850: -(1-lambda)*PG/sum(mask) + lambda*model_loss, the sampled
    rows and the ground-truth rows in one teacher-forced pass, vs float64 autograd with the same dropout masks."""
    import torch
    import s2vt_amd
    from s2vt_amd import model as M
    from oracle import s2vt_torch as T
    case = CASES[1]
    d, p, video, cap, vid, sid, N = _setup(oracle, case)
    B, rep, keep, lam = case["B"], case["rep"], 0.9, 0.5
    rng = np.random.default_rng(12)
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    r = rng.random(N).astype(np.float32) * 2; b = np.tile(rng.random(B).astype(np.float32) * 2, rep)
    gcap = rng.integers(0, d.n_words, (B, d.n_caption_lstm_step)).astype(np.int32); gcap[:, -3:] = 0
    gmask = s2vt_amd.hostglue.masks_from_ids(gcap)
    mdl = M.Video_Caption_Generator(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, B, 0, d.n_video_lstm_step,
                                    d.n_caption_lstm_step, dropout_rate=keep)
    mdl.store.load(p)
    s1 = mdl.dropout_seed + 104729 * mdl.global_step
    drop1 = oracle.dropout_masks(s1, vid, sid, keep, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    # the ground-truth rows ride in the same pass as "sample" number rep of each video: their own dropout masks
    drop2 = oracle.dropout_masks(s1, vid[:B], np.full(B, rep, np.int32), keep, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    pt = T.to_torch(p, torch.float64, True)
    lg1 = T.teacher_forced(pt, torch.as_tensor(np.tile(video, (rep, 1, 1))).double(), cap, drop1, keep)
    lg2 = T.teacher_forced(pt, torch.as_tensor(video).double(), gcap, drop2, keep)
    ref = (1 - lam) * T.pg_loss(lg1, cap, mask, r, b) + lam * T.xe_loss(pt, lg2, gcap, gmask, q1=True)
    ref.backward()
    st = mdl.mixed_update(video, cap, mask, r, b, gcap, gmask, lr=0.0, lambda_loss=lam, video_base=5)
    wd = sum(0.5 * float((mdl.store.p[n].double() ** 2).sum()) for n in mdl.store.names if n not in M.UNDECAYED)
    assert abs(float(st.loss) + lam * mdl.decay_value * wd - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    for n in mdl.store.names:
        g = mdl.store.g[n].cpu().numpy().astype(np.float64)
        rg = pt[n].grad.numpy()
        assert np.abs(g - rg).max() <= 2e-4 * (np.abs(rg).max() + 1e-12) + 1e-9, n


def test_backward_phases_sum_to_the_whole_pass(gpu, oracle):
    """s2vt_bptt_bwd_phase 1, 3, 4 (the data-parallel split: each slice's all-reduce starts when its phase ends) leaves
    the same gradients as the single call; after phase 1 only the vocab projection is touched, after 3 also LSTM2."""
    import torch
    case = CASES[1]
    d, p, video, cap, vid, sid, N = _setup(oracle, case)
    dims = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    dp_ = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp_)
    rng = np.random.default_rng(2)
    coef = _dev(rng.standard_normal(N * d.n_caption_lstm_step).astype(np.float32))

    def run(phases):
        logits, ws = gpu.teacher_forced_fwd(dims, params, _dev(video), _dev(cap), N, 0.9, 99, _dev(vid), _dev(sid))
        gpu.softmax_nll_fwd_bwd(logits, _dev(cap).t().contiguous().view(-1), coef, 0.0)
        g = {k: torch.zeros_like(v) for k, v in dp_.items()}
        grads = gpu.make_params(g)
        snaps = []
        for ph in phases:
            gpu.bptt_bwd(dims, params, grads, _dev(video), N, logits, ws, 0.9, 99, _dev(vid), _dev(sid), phase=ph)
            snaps.append({k: v.clone() for k, v in g.items()})
        return g, snaps
    whole, _ = run([0])
    parts, snaps = run([1, 3, 4])
    for k in whole:
        ref = whole[k].cpu().numpy(); got = parts[k].cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-12) + 1e-9, k
    touched1 = {k for k, v in snaps[0].items() if float(v.abs().sum()) > 0}
    assert touched1 == {"embed_word_W", "embed_word_b"}
    touched3 = {k for k, v in snaps[1].items() if float(v.abs().sum()) > 0}
    assert touched3 == {"embed_word_W", "embed_word_b", "lstm2_W", "lstm2_b"}
    for k in ("lstm2_W", "lstm2_b", "embed_word_W"):
        assert torch.equal(snaps[1][k], parts[k])                 # final when their phase returns


def test_device_glue_matches_host_helpers(gpu):
    """s2vt_caption_mask / s2vt_pg_coef / s2vt_step_scalars against the host helpers pinned to the reference
    (hostglue.masks_from_ids <-> decode_captions_masks, tests/golden/hostglue.json) and plain numpy; and reinforce_update with
    mask=None (library glue) lands where the explicit-mask form does."""
    import torch
    from s2vt_amd import hostglue, model as M
    rng = np.random.default_rng(5)
    N, Tc = 37, 9
    ids = rng.integers(0, 6, (N, Tc)).astype(np.int32); ids[3] = 5; ids[4, 0] = 0          # a row without <eos>, a row that starts with it
    tail = torch.zeros(1, device="cuda")
    mask, tgt, msum = gpu.caption_mask(torch.as_tensor(ids).cuda(), mask_sum_copy=tail)
    ref = hostglue.masks_from_ids(ids)
    assert np.array_equal(mask.cpu().numpy(), ref) and np.array_equal(tgt.cpu().numpy(), ids.T.reshape(-1))
    assert float(msum) == float(ref.sum()) == float(tail)
    r = rng.random(N).astype(np.float32); b = rng.random(N).astype(np.float32)
    coef = gpu.pg_coef(mask, torch.as_tensor(r).cuda(), torch.as_tensor(b).cuda(), 0.75)
    assert np.array_equal(coef.cpu().numpy(), (ref * ((r - b) * np.float32(0.75))[:, None]).T.reshape(-1))
    nll = torch.as_tensor(rng.random(N * Tc).astype(np.float32)).cuda()
    loss = torch.empty(1, device="cuda"); gs = torch.empty(1, device="cuda"); sq = torch.ones(1, device="cuda")
    g = torch.full((1,), 8.0, device="cuda")
    gpu.step_scalars(coef, nll, msum, g, loss, gs, sq)
    want = float((coef.double() * nll.double()).sum() / float(ref.sum()))
    assert abs(float(loss) - want) <= 1e-5 * max(1.0, abs(want)) and float(gs) == 0.125 and float(sq) == 0.0
    # the fused form of the update == the explicit-mask form
    video = torch.as_tensor(np.abs(rng.standard_normal((4, 5, 128)) * 0.5).astype(np.float32)).cuda()
    outs = []
    for fused in (False, True):
        mdl = M.Video_Caption_Generator(128, 260, 32, 64, 4, 0, 5, 8, seed=5, dropout_rate=0.9)
        s, _ = mdl.sample(video, 3, True, seed=9)
        mk = None if fused else torch.as_tensor(hostglue.masks_from_ids(s.cpu().numpy())).cuda()
        rr = (rng.random(12) * 0 + np.linspace(0.1, 2, 12)).astype(np.float32); bb = np.tile(np.float32([0.5, 1.0, 0.2, 0.9]), 3)
        st = mdl.reinforce_update(video, s, mk, rr, bb, lr=1e-3, reuse_sampler_state=True)
        outs.append((mdl.store.theta.cpu().numpy(), float(st.loss), float(st.grad_sumsq), float(st.mask_sum)))
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 2e-4
    assert abs(outs[0][1] - outs[1][1]) <= 1e-5 * max(1.0, abs(outs[0][1])) and outs[0][3] == outs[1][3]
    assert abs(outs[0][2] - outs[1][2]) <= 1e-4 * outs[0][2]
