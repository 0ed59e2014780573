"""GPU: the CNN-in-the-loop seam (e2e.py, s2vt_bptt_dvideo) -- SURVEY 8(f) rank 3.

The hand-written half is checked the usual way: d(loss)/d(video) from the library against float64 autograd through
the torch restatement with the same dropout masks.  The wiring (one global-norm clip and one TF-form Adam over CNN +
captioner, weight decay on every variable, attribute-head gradient into the features) is checked with a small
stand-in CNN so the suite does not pay Inception-ResNet-v2's MIOpen warm-up; tools/e2e_irv2_smoke.py runs the real one.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tiny_cnn(D, seed=0):
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    return nn.Sequential(nn.Conv2d(3, 8, 3, stride=2), nn.ReLU(), nn.Conv2d(8, 8, 3), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, D), nn.ReLU())


def _case(oracle, label_dim=0):
    d = oracle.Dims(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6, label_dim=label_dim)
    p = oracle.init_params(d, seed=3, attr=label_dim > 0)
    rng = np.random.default_rng(8)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    return d, p, rng


@pytest.mark.parametrize("rep", [1, 3])
def test_video_grad_vs_float64_autograd(gpu, oracle, rep):
    import torch
    import s2vt_amd
    from s2vt_amd import model as M
    from oracle import s2vt_torch as T
    d, p, rng = _case(oracle)
    B, keep = 4, 0.9
    N = B * rep
    video = np.abs(rng.standard_normal((B, 3, 24)) * 0.5).astype(np.float32)
    cap = rng.integers(0, 97, (N, 6)).astype(np.int32); cap[:, -1] = 0; cap[1, 2] = 0
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    r = rng.random(N).astype(np.float32); b = np.tile(rng.random(B).astype(np.float32), rep)
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, B, 0, 3, 6, dropout_rate=keep)
    mdl.store.load(p)
    vid = np.tile(np.arange(B, dtype=np.int32), rep); sid = np.repeat(np.arange(rep, dtype=np.int32), B)
    drop = oracle.dropout_masks(mdl.dropout_seed, vid, sid, keep, 20, 3, 6)
    pt = T.to_torch(p, torch.float64, True)
    vt = torch.as_tensor(video).double().requires_grad_()
    loss = T.pg_loss(T.teacher_forced(pt, vt.repeat(rep, 1, 1), cap, drop, keep), cap, mask, r, b)
    loss.backward()
    got = {}

    def hook(gscale):
        got["dv"] = (mdl.video_grad() * gscale).cpu().numpy()
        return torch.zeros(1, device="cuda")
    mdl.reinforce_update(video, cap, mask, r, b, lr=0.0, extra_sumsq=hook)
    ref = vt.grad.numpy()
    assert got["dv"].shape == ref.shape
    assert np.abs(got["dv"] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-10


def test_e2e_xe_step_wiring_vs_float64_autograd(gpu, oracle):
    """e2e_tf_s2vt.py train(): XE through the CNN, decay on every variable, clip 10 on the joint norm, one Adam."""
    import torch
    import s2vt_amd
    from s2vt_amd import e2e, model as M
    from oracle import s2vt_torch as T
    d, p, rng = _case(oracle)
    B, keep = 4, 0.9
    frames = rng.uniform(-1, 1, (B, 3, 3, 17, 17)).astype(np.float32)
    cap = rng.integers(0, 97, (B, 6)).astype(np.int32); cap[:, -2:] = 0
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, B, 0, 3, 6, dropout_rate=keep)
    mdl.store.load(p)
    cnn = _tiny_cnn(24)
    ref_cnn = copy.deepcopy(cnn).double()
    tr = e2e.EndToEnd(mdl, cnn, feature_keep=1.0)
    # float64 reference of the whole graph
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    drop = oracle.dropout_masks(mdl.dropout_seed, vid, sid, keep, 20, 3, 6)
    pt = T.to_torch(p, torch.float64, True)
    feats = ref_cnn(torch.as_tensor(frames).double().reshape(B * 3, 3, 17, 17)).reshape(B, 3, 24)
    loss = T.xe_loss(pt, T.teacher_forced(pt, feats, cap, drop, keep), cap, mask, q1=True)       # decays the non-'bias' captioner variables
    extra = sum(0.5 * (pt[n] ** 2).sum() for n in M.UNDECAYED) + sum(0.5 * (q ** 2).sum() for q in ref_cnn.parameters())
    (loss + mdl.decay_value * extra).backward()
    theta0 = tr.theta.clone()
    st = tr.xe_step(torch.as_tensor(frames), cap, mask, lr=0.0)
    assert torch.equal(tr.theta, theta0)                                    # lr 0: gradients only
    ref_flat = np.concatenate([q.grad.numpy().ravel() for q in ref_cnn.parameters()])
    got = tr.grad.cpu().numpy()
    assert np.abs(got - ref_flat).max() <= 3e-4 * np.abs(ref_flat).max() + 1e-10
    for n in mdl.store.names:
        rg = pt[n].grad.numpy()
        assert np.abs(mdl.store.g[n].cpu().numpy() - rg).max() <= 3e-4 * (np.abs(rg).max() + 1e-12) + 1e-9, n
    total = sum(float((pt[n].grad ** 2).sum()) for n in mdl.store.names) + float((ref_flat ** 2).sum())
    assert abs(float(st.grad_sumsq) - total) <= 1e-3 * total               # ONE global norm over both halves
    # the views are live: module parameters alias the flat buffer, and steps with lr > 0 move both halves and learn
    w0 = mdl.store.p["lstm1_W"].clone()
    losses = [float(tr.xe_step(torch.as_tensor(frames), cap, mask, lr=2e-2).loss) for _ in range(12)]
    assert not torch.equal(tr.theta, theta0) and not torch.equal(mdl.store.p["lstm1_W"], w0)
    assert next(cnn.parameters()).data_ptr() == tr.theta.data_ptr()
    assert losses[-1] < losses[0]
    # clipped joint update: the Adam step of the CNN half used the same clip factor as the captioner's (TF form)
    assert mdl.global_step == 13


def test_e2e_xe_step_with_frozen_cnn(gpu, oracle):
    """fix_e2e_tf_s2vt.py (:120, :284): the CNN in the loop behind tf.stop_gradient.  No data gradient reaches it, but the script's
    weight-decay term sums over every trainable variable (:199, always-true predicate) and compute_gradients(tf_loss) covers them
    (:534-535): the CNN variables get decay_value * theta, count toward the joint clip norm and are moved by Adam.  The captioner sees
    the same features and gets the gradients model.xe_update gives on them.  cnn_weight_decay=False is the literal freeze."""
    import torch
    import s2vt_amd
    from s2vt_amd import e2e, model as M
    d, p, rng = _case(oracle)
    B, keep = 4, 0.9
    frames = torch.as_tensor(rng.uniform(-1, 1, (B, 3, 3, 17, 17)).astype(np.float32))
    cap = rng.integers(0, 97, (B, 6)).astype(np.int32); cap[:, -2:] = 0
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    outs = []
    for mode in ("frozen", "literal", "captioner"):
        mdl = M.Video_Caption_Generator(24, 97, 12, 20, B, 0, 3, 6, dropout_rate=keep)
        mdl.store.load(p)
        tr = e2e.EndToEnd(mdl, _tiny_cnn(24), feature_keep=keep, seed=11)
        theta0 = tr.theta.clone()
        if mode == "frozen":
            st = tr.xe_step(frames, cap, mask, lr=0.0, freeze_cnn=True)
            # the CNN half of the gradient list is decay * theta, nothing else, and it is part of the ONE global norm
            assert torch.equal(tr.grad, theta0 * mdl.decay_value)
            cnn_sq = float((theta0.double() * mdl.decay_value).pow(2).sum())
        elif mode == "literal":
            st = tr.xe_step(frames, cap, mask, lr=0.0, freeze_cnn=True, cnn_weight_decay=False)
        else:                                                                                   # what the captioner alone does on the same features
            video, _ = tr.extract(frames, dropout=True, track=False)
            st = mdl.xe_update(video, cap, mask, 0.0, decay_all=True)
        outs.append((float(st.loss), float(st.grad_sumsq), mdl.store.grad[:mdl.store.numel].clone()))
        if mode != "captioner":                                                                 # and a real step
            w0 = mdl.store.p["lstm1_W"].clone()
            tr.xe_step(frames, cap, mask, lr=1e-2, freeze_cnn=True, cnn_weight_decay=(mode == "frozen"))
            assert not torch.equal(mdl.store.p["lstm1_W"], w0)
            if mode == "frozen":
                # TF-form Adam on the CNN half, two updates with g = decay * theta0 (the first at lr = 0 moved only the moments; the joint
                # norm is far below the clip): its own update count (2), not the captioner's
                g = theta0.double() * mdl.decay_value
                m2, v2 = 0.19 * g, (1 - 0.999 ** 2) * g * g
                want = theta0.double() - 1e-2 * (1 - 0.999 ** 2) ** 0.5 / (1 - 0.9 ** 2) * m2 / (v2.sqrt() + 1e-8)
                assert tr.adam_t == 2 and float((tr.theta.double() - want).abs().max()) <= 1e-5 * 1e-2 + 1e-7
                assert float((tr.theta - theta0).abs().max()) > 1e-3
            else:
                assert torch.equal(tr.theta, theta0) and float(tr.grad.abs().max()) == 0.0 and tr.adam_t == 0
    assert outs[0][0] == outs[1][0] == outs[2][0]                                               # the same forward, bit for bit
    assert abs(outs[1][1] - outs[2][1]) <= 1e-5 * outs[2][1]                                    # (gradients: order-free reductions)
    assert abs(outs[0][1] - (outs[2][1] + cnn_sq)) <= 1e-5 * outs[0][1]                         # joint norm = captioner's + ||decay theta_cnn||^2
    for k in (0, 1):
        assert float((outs[k][2] - outs[2][2]).abs().max()) <= 1e-5 * float(outs[2][2].abs().max())


def test_e2e_reinforce_multitask_step(gpu, oracle):
    """reinforce_multitask_e2e_attribute_loss.py:957 through the CNN: PG + attribute BCE; the attribute head's
    gradient reaches the features as well."""
    import torch
    import s2vt_amd
    from s2vt_amd import e2e, model as M
    from oracle import s2vt_torch as T
    d, p, rng = _case(oracle, label_dim=10)
    B, K, keep, alpha = 4, 2, 0.9, 0.3
    frames = rng.uniform(-1, 1, (B, 3, 3, 17, 17)).astype(np.float32)
    y = (rng.random((B, 10)) < .3).astype(np.float32)
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, B, 0, 3, 6, dropout_rate=keep, label_dim=10, alpha=alpha)
    mdl.store.load(p)
    cnn = _tiny_cnn(24, seed=1)
    ref_cnn = copy.deepcopy(cnn).double()
    tr = e2e.EndToEnd(mdl, cnn, feature_keep=1.0)
    rb = {}

    def reward_fn(samples, greedy):
        rb["r"] = rng.random(samples.shape[0]).astype(np.float32); rb["b"] = rng.random(greedy.shape[0]).astype(np.float32)
        return rb["r"], rb["b"]
    st = tr.reinforce_step(torch.as_tensor(frames), reward_fn, lr=0.0, K=K, true_labels=y, sample_seed=11)
    samples = st.samples.cpu().numpy()
    assert samples.shape == (K * B, 6) and st.greedy.shape == (B, 6)
    mask = s2vt_amd.hostglue.masks_from_ids(samples)
    vid = np.tile(np.arange(B, dtype=np.int32), K); sid = np.repeat(np.arange(K, dtype=np.int32), B)
    drop = oracle.dropout_masks(mdl.dropout_seed, vid, sid, keep, 20, 3, 6)          # the step ran at global_step 0
    pt = T.to_torch(p, torch.float64, True)
    feats = ref_cnn(torch.as_tensor(frames).double().reshape(B * 3, 3, 17, 17)).reshape(B, 3, 24)
    lg = T.teacher_forced(pt, feats.repeat(K, 1, 1), samples, drop, keep)
    loss = (1 - alpha) * T.pg_loss(lg, samples, mask, rb["r"], np.tile(rb["b"], K)) + alpha * T.attr_bce(pt, feats, y)
    loss.backward()
    ref_flat = np.concatenate([q.grad.numpy().ravel() for q in ref_cnn.parameters()])
    got = tr.grad.cpu().numpy()
    assert np.abs(got - ref_flat).max() <= 3e-4 * np.abs(ref_flat).max() + 1e-10
    # greedy ids through generate() equal the step's greedy pass (feature dropout off in both)
    assert torch.equal(tr.generate(torch.as_tensor(frames)), st.greedy)


def test_feature_dropout_is_slim_dropout(gpu, oracle):
    import torch
    from s2vt_amd import e2e, model as M
    d, p, rng = _case(oracle)
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, 4, 0, 3, 6, dropout_rate=0.9)
    mdl.store.load(p)
    tr = e2e.EndToEnd(mdl, torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(3, 24)))
    frames = torch.as_tensor(rng.uniform(-1, 1, (64, 3, 3, 9, 9)).astype(np.float32))
    clean, _ = tr.extract(frames, dropout=False)
    dropped, _ = tr.extract(frames, dropout=True)
    kept = dropped != 0
    assert 0.85 < float(kept.float().mean()) < 0.95
    assert torch.allclose(dropped[kept], clean[kept] / 0.9, rtol=1e-6, atol=0)       # kept units scaled by 1/keep_prob
    again, _ = tr.extract(frames, dropout=True)
    assert torch.equal(again, dropped)                                               # counter-based on (seed, global_step)
    # the mask is the library's Philox dropout stream keyed by (GLOBAL video, frame, unit): the oracle reproduces it, ...
    vid = np.repeat(np.arange(64, dtype=np.int32), 3); frame = np.tile(np.arange(3, dtype=np.int32), 64)
    seed = tr.seed + 15485863 * (mdl.global_step + 1)
    ref_mask = oracle.dropout_mask(seed, vid, frame, 768, 0.9, 24).reshape(64, 3, 24)
    assert np.array_equal(kept.cpu().numpy(), (ref_mask != 0) & (clean.cpu().numpy() != 0))
    # ... a rank holding videos [16, 24) of the global batch applies the very masks the one-rank run applies to them, ...
    shard, _ = tr.extract(frames[16:24], dropout=True, video_base=16)
    assert torch.equal(shard, dropped[16:24])
    # ... and independent draws (the sampler graph's and the loss graph's dropout ops) get independent masks from one CNN pass
    (d0, _), (d1, _) = tr.extract(frames, dropout=True, draws=(0, 1))
    assert torch.equal(d0, dropped) and not torch.equal(d1, dropped) and 0.85 < float((d1 != 0).float().mean()) < 0.95


def test_e2e_real_inception_resnet_v2_step(gpu):
    """BASELINE configs[4] with the REAL network (irv2.InceptionResnetV2, 244 conv units, MIOpen convolutions) in front of
    the HIP captioner at the reference's dimensions (299 x 299 frames, d = 1536, E = 500, H = 1000, Tc = 20; batch 2 x 5
    frames, a 2,000-word vocabulary to keep it light).  No TF checkpoint exists to compare activations with (the layer
    table, paddings and BN formula are pinned on the CPU side, tests/test_irv2_cpu.py), so this checks the SEAM on the real
    network: the features the captioner sees are the network's pooled output with slim.dropout applied, the gradient that
    reaches the CNN is the directional derivative of the captioner's loss w.r.t. those features (a finite-difference probe
    along the gradient itself), one global norm spans both halves, and steps with lr > 0 reduce the loss."""
    import torch
    from s2vt_amd import e2e, hostglue, irv2, model as M
    torch.manual_seed(0)
    B, TV, TC, D, E, H, V = 2, 5, 20, 1536, 500, 1000, 2000
    cnn = irv2.InceptionResnetV2()
    mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, TV, TC, dropout_rate=1.0, seed=3)      # no LSTM dropout: the probe below re-runs the loss
    tr = e2e.EndToEnd(mdl, cnn, feature_keep=1.0)
    rng = np.random.default_rng(0)
    frames = torch.as_tensor(rng.uniform(-1, 1, (B, TV, 3, 299, 299)).astype(np.float32))
    cap = rng.integers(2, V, (B, TC)).astype(np.int32); cap[0, 7:] = 0; cap[1, 12:] = 0
    mask = hostglue.masks_from_ids(cap)
    feats, _ = tr.extract(frames, dropout=False)
    assert feats.shape == (B, TV, D) and float(feats.min()) >= 0.0 and bool(torch.isfinite(feats).all())
    with torch.no_grad():
        direct = cnn(frames.cuda().reshape(B * TV, 3, 299, 299)).reshape(B, TV, D)
    assert torch.allclose(feats, direct, rtol=1e-4, atol=1e-6)          # (MIOpen may pick a different algorithm per call: not bitwise)
    # lr = 0 step: gradients only
    st = tr.xe_step(frames, cap, mask, lr=0.0)
    g_cnn = tr.grad.clone()
    assert bool(torch.isfinite(g_cnn).all()) and float(g_cnn.abs().max()) > 0
    cap_sq = sum(float((mdl.store.g[n].double() ** 2).sum()) for n in mdl.store.names)
    cnn_sq = float((g_cnn.double() ** 2).sum())
    assert abs(float(st.grad_sumsq) - (cap_sq + cnn_sq)) <= 1e-3 * (cap_sq + cnn_sq)       # ONE clip norm over both halves
    # the gradient handed across the seam: d loss / d features, probed by a finite difference along itself
    mdl.global_step = 0
    dv = mdl.video_grad() / float(mask.sum())                                              # (unnormalised in the workspace)

    def xe_loss_of(f):
        mdl.global_step = 0
        c = torch.as_tensor(cap).cuda(); m = torch.as_tensor(mask).cuda()
        coef = ((m.sum(0)[:, None] / float(B)).expand(-1, B)).contiguous().view(-1)
        nll, _ = mdl._forward_loss(f.contiguous(), c, coef, 0.05, 1, 0, 1.0)
        return float(torch.dot(coef, nll).double() / m.sum().double())
    eps = 1e-2 / float(dv.norm())
    num = (xe_loss_of(feats + eps * dv) - xe_loss_of(feats - eps * dv)) / (2 * eps)
    assert abs(num - float((dv.double() ** 2).sum())) <= 5e-2 * float((dv.double() ** 2).sum())
    # and it learns through the real network
    mdl.global_step = 0
    losses = [float(tr.xe_step(frames, cap, mask, lr=1e-3).loss) for _ in range(4)]
    assert losses[-1] < losses[0] and np.isfinite(losses).all()


def test_config4_per_gpu_shape_xe_and_reinforce_steps(gpu):
    """BASELINE configs[4] at its per-GPU size -- B = 16 (128 over 8 GPUs) x 5 frames x 3 x 299 x 299 through the real
    Inception-ResNet-v2, d = 1536, E = 500, H = 1000, Tc = 20, |V| = 12000 -- one XE step (e2e_tf_s2vt.py:482-700) and one
    REINFORCE step (reinforcement_e2e.py:1085-1140).  Seam checks as in test_e2e_real_inception_resnet_v2_step: the
    captioner's input is the CNN's pooled output under the step's slim.dropout mask, ONE global norm spans both halves, the
    update moves both halves, the sampler's ids are the captioner's own on those features; no persistent-recurrence fault."""
    import torch
    from s2vt_amd import e2e, hostglue, irv2, model as M, ops
    torch.manual_seed(0)
    B, TV, TC, D, E, H, V = 16, 5, 20, 1536, 500, 1000, 12000
    cnn = irv2.InceptionResnetV2()
    mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, TV, TC, dropout_rate=0.9, seed=3)
    tr = e2e.EndToEnd(mdl, cnn, seed=5)
    rng = np.random.default_rng(0)
    frames = torch.as_tensor(rng.uniform(-1, 1, (B, TV, 3, 299, 299)).astype(np.float32)).cuda()
    ln = 1 + np.minimum(rng.poisson(6, B), TC - 2)
    cap = rng.integers(2, V, (B, TC)).astype(np.int32)
    for j in range(B):
        cap[j, ln[j]:] = 0
    mask = hostglue.masks_from_ids(cap)
    # ---- the seam, forward: features with the step's dropout mask == pooled output x mask / keep, rows zeroed ~10 %
    feats, _ = tr.extract(frames, dropout=True)
    clean, _ = tr.extract(frames, dropout=False)
    assert feats.shape == (B, TV, D) and bool(torch.isfinite(feats).all()) and float(clean.min()) >= 0.0
    kept = feats != 0
    assert torch.allclose(feats[kept], (clean / 0.9)[kept], rtol=2e-4, atol=1e-6)
    assert 0.85 < float(kept.float().mean() / (clean != 0).float().mean()) < 0.95
    # ---- XE step, lr 0: gradients of both halves, one norm
    theta_c, theta_s = tr.theta.clone(), mdl.store.theta.clone()
    st = tr.xe_step(frames, cap, mask, lr=0.0)
    cap_sq = sum(float((mdl.store.g[n].double() ** 2).sum()) for n in mdl.store.names)
    cnn_sq = float((tr.grad.double() ** 2).sum())
    assert cnn_sq > 0 and np.isfinite(cnn_sq) and np.isfinite(float(st.loss))
    assert abs(float(st.grad_sumsq) - (cap_sq + cnn_sq)) <= 1e-3 * (cap_sq + cnn_sq)
    assert 8.0 < float(st.loss) < 11.0                        # ~ log(12000) = 9.39 at initialisation
    # ---- XE step with the reference's learning rate: both halves move, by at most ~lr per entry (Adam)
    st = tr.xe_step(frames, cap, mask, lr=1e-5)
    for new, old in ((tr.theta, theta_c), (mdl.store.theta, theta_s)):
        d = (new - old).abs().max()
        assert 0 < float(d) <= 2.5e-5
    # ---- REINFORCE step: sample (dropout on) + greedy (off) through the CNN, PG update through the CNN
    r = rng.random(B).astype(np.float32); b = rng.random(B).astype(np.float32)
    theta_c = tr.theta.clone()
    st = tr.reinforce_step(frames, lambda s_, g_: (r, b), lr=1e-6, K=1, sample_seed=77)
    assert st.samples.shape == (B, TC) and st.greedy.shape == (B, TC) and np.isfinite(float(st.loss))
    assert int(st.samples.min()) >= 0 and int(st.samples.max()) < V
    assert 0 < float((tr.theta - theta_c).abs().max()) <= 2.5e-6
    torch.cuda.synchronize()
    assert ops.chain_timeouts() == 0 and not ops.chain_fault()
    g = tr.generate(frames)                                   # build_generator through the updated network
    assert g.shape == (B, TC) and int(g.min()) >= 0 and int(g.max()) < V
