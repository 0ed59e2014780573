"""The reference script's OWN default configuration as a workload of its own: reinforcement_multisampling_tf_s2vt.py:505-517,743-753
-- batch 256, K = 8 samples per video, Tc = 35, |V| = 9,972 (Tv = 5, d = 1536, E = 500, H = 1000): 2304 sampler rows, N = 2048 update
rows (beyond every persistent form: the recurrences run as per-step launches), 2.9 GB of logits per pass, operands beyond 2 GiB.
`bench.py --workload rl_ref` times it.  The oracle cannot run 2048 rows x 35 steps x 9972 words in a test's time, so parity at this
size is (a) oracle spot rows -- rows of different videos never interact, so the sampled / greedy ids and the teacher-forced logits of two
videos of the batch must equal the oracle run on those two videos alone -- and (b) size-independent properties of the update: linearity in
(r - b), dlogits rows summing to zero, the mask normaliser, and shard additivity (the unnormalised gradient of the 256 videos = the sum
over two shards of 128, which is also the data-parallel contract of SURVEY 8(e) at this size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B, K, TC, TV, D, E, H, V = 256, 8, 35, 5, 1536, 500, 1000, 9972


@pytest.fixture(scope="module")
def ref(gpu):
    import torch
    from s2vt_amd import model as M
    mdl = M.Video_Caption_Generator(D, V, E, H, B, TV + TC, TV, TC, seed=77, multisample=K)
    rng = np.random.default_rng(5)
    for n in ("lstm1_b", "lstm2_b", "encode_image_b"):
        mdl.store.p[n].copy_(torch.as_tensor(rng.uniform(-.1, .1, mdl.store.shapes[n]).astype(np.float32)))
    # (a trained model's <eos> bias: samples end, so the masks are ragged as in the reference's loop -- with zero biases a
    #  random-initialised policy never emits <eos> and every mask would be full)
    b = rng.uniform(-.5, .5, V).astype(np.float32); b[0] = 6.5
    mdl.store.p["embed_word_b"].copy_(torch.as_tensor(b))
    g = torch.Generator().manual_seed(9)
    video = (torch.randn(B, TV, D, generator=g) * 0.5).abs().cuda()
    yield mdl, video
    del mdl
    torch.cuda.empty_cache()


def test_reference_config_sampler_spot_rows_vs_oracle(ref, oracle):
    import torch
    mdl, video = ref
    s, g = mdl.sample(video, K, True, seed=31)
    assert s.shape == (K * B, TC) and g.shape == (B, TC) and int(s.min()) >= 0 and int(s.max()) < V
    s2, g2 = mdl.sample(video, K, True, seed=31)
    assert torch.equal(s, s2) and torch.equal(g, g2)
    p = {n: mdl.store.p[n].cpu().numpy() for n in mdl.store.names}
    d = oracle.Dims(D, V, E, H, TV, TC, 0)
    for lo in (0, 201):                                             # first tile and a row tile deep in the 2304-row launches
        rs, rg = oracle.sample_captions(p, d, video[lo:lo + 2].cpu().numpy(), K, seed=31, video_base=lo)
        assert np.array_equal(s.view(K, B, TC)[:, lo:lo + 2].reshape(-1, TC).cpu().numpy(), rs)
        assert np.array_equal(g[lo:lo + 2].cpu().numpy(), rg)
    ends = (s == 0).any(1).float().mean()
    assert 0.5 < float(ends) <= 1.0                                 # the masks below are ragged


def test_reference_config_teacher_forced_logits_spot_rows_vs_oracle(ref, oracle, gpu):
    import torch
    mdl, video = ref
    s, _ = mdl.sample(video, K, False, seed=32)
    N = K * B
    vid, sid = mdl._row_ids(B, K, 0)
    seed = 4242
    logits, _ = gpu.teacher_forced_fwd(mdl.dims, mdl.store.params, video, s, N, 0.9, seed, vid, sid)
    assert logits.shape == (TC * N, V)                              # 2.86 GB: the vector path addresses per tile, not from the matrix base
    p = {n: mdl.store.p[n].cpu().numpy() for n in mdl.store.names}
    d = oracle.Dims(D, V, E, H, TV, TC, 0)
    lg = logits.view(TC, N, V)
    for j in (3, 255):
        rows = np.arange(K) * B + j                                 # the K samples of video j (sample-major rows k*B + j)
        cap = s[torch.as_tensor(rows).cuda()].cpu().numpy().astype(np.int32)
        drop = oracle.dropout_masks(seed, np.full(K, j, np.int32), np.arange(K, dtype=np.int32), 0.9, H, TV, TC)
        ref_l = oracle.teacher_forced(p, d, np.tile(video[j:j + 1].cpu().numpy(), (K, 1, 1)), cap, drop, 0.9)
        got = lg[:, torch.as_tensor(rows).cuda()].permute(1, 0, 2).cpu().numpy()
        assert np.array_equal(got, ref_l), j
    del logits, lg
    torch.cuda.empty_cache()


def test_reference_config_update_properties(ref, gpu):
    """One whole REINFORCE update at N = 2048 (lr = 0): linearity in (r - b), dlogits rows sum to zero, the mask normaliser, and shard
    additivity -- g(256 videos) * sum(mask) == g(videos 0..127) * sum(mask_a) + g(videos 128..255) * sum(mask_b) with the global video
    indices in the noise counters (video_base), i.e. two ranks x 128 == one rank x 256 before the all-reduce."""
    import torch
    from s2vt_amd import hostglue
    mdl, video = ref
    s, _ = mdl.sample(video, K, True, seed=33)
    N = K * B
    mask_h = hostglue.masks_from_ids(s.cpu().numpy())
    assert 0.05 < mask_h.mean() < 0.9
    mask = torch.as_tensor(mask_h).cuda()
    g = torch.Generator().manual_seed(2)
    r = (torch.rand(N, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(K).cuda()
    step0 = mdl.global_step
    gpu.prof_filter(-1, -1); gpu.prof_enable(True)
    st1 = mdl.reinforce_update(video, s, mask, r, b, lr=0.0, clip_norm=5.0, reuse_sampler_state=True)
    torch.cuda.synchronize()
    rows = gpu.prof_collect(); gpu.prof_enable(False)
    classes = {r_["kernel_class"] for r_ in rows}
    assert 1 in classes, classes                                    # 2048 rows: LSTM2's recurrence runs as per-step cell launches (LSTM1's 256 rows may be persistent)
    assert (3, "tn128x128(dma)") in {(r_["kernel_class"], r_["name"]) for r_ in rows}
    g1 = mdl.store.grad[:mdl.store.numel].clone()
    dl = mdl._ctx[2]
    assert float(dl.sum(1).abs().max()) < 1e-4
    assert float(st1.mask_sum) == float(mask_h.sum()) and float(mask[:, 0].min()) == 1.0
    assert np.isfinite(float(st1.loss)) and float(st1.grad_sumsq) > 0
    del dl
    mdl.global_step = step0                                                     # same dropout masks
    st2 = mdl.reinforce_update(video, s, mask, 2 * r, 2 * b, lr=0.0, clip_norm=5.0)   # (and without the sampler-state reuse: the same update)
    g2 = mdl.store.grad[:mdl.store.numel]
    assert torch.allclose(g2, 2 * g1, rtol=1e-3, atol=1e-6 * float(g1.abs().max()) + 1e-12)
    assert abs(float(st2.loss) - 2 * float(st1.loss)) < 1e-4 * abs(float(st1.loss)) + 1e-6
    assert abs(float(st2.grad_sumsq) - 4 * float(st1.grad_sumsq)) < 1e-2 * float(st1.grad_sumsq)
    # the host-mask path (active steps + live rows, as train_rl has it) gives the same update
    mdl.global_step = step0
    st3 = mdl.reinforce_update(video, s, mask_h, r, b, lr=0.0, clip_norm=5.0)
    assert mdl._ctx[8] == mdl.active_steps(mask_h)
    g3 = mdl.store.grad[:mdl.store.numel]
    assert abs(float(st3.loss) - float(st1.loss)) <= 2e-6 * max(1.0, abs(float(st1.loss)))
    assert float((g3 - g1).abs().max()) <= 3e-5 * float(g1.abs().max())
    # shard additivity
    acc = torch.zeros_like(g1, dtype=torch.float64)
    half = B // 2
    s3 = s.view(K, B, TC); m3 = mask.view(K, B, TC); r3 = r.view(K, B); b3 = b.view(K, B)
    for lo in (0, half):
        mdl.global_step = step0
        sl = slice(lo, lo + half)
        st_h = mdl.reinforce_update(video[sl].contiguous(), s3[:, sl].reshape(-1, TC).contiguous(), m3[:, sl].reshape(-1, TC).contiguous(),
                                    r3[:, sl].reshape(-1).contiguous(), b3[:, sl].reshape(-1).contiguous(), lr=0.0, clip_norm=5.0, video_base=lo)
        acc += mdl.store.grad[:mdl.store.numel].double() * float(st_h.mask_sum)
    mdl.global_step = step0
    want = g1.double() * float(st1.mask_sum)
    assert float((acc - want).abs().max()) <= 3e-5 * float(want.abs().max())


def test_reference_config_gradients_vs_float64_autograd_on_a_row_slice(ref, oracle):
    """VERDICT r5 missing 4: the gradient TENSORS of the reference-default update (N = 2048 rows x 35 steps x 9,972 words: the big-M tile
    table, per-step LSTM2 recurrences, 2.9 GB of logits) against float64 autograd.  The whole update runs on the GPU at full size, dense (the
    mask lives on the device: no truncated unroll, no live-row packing); the mask is zero outside ONE 256-row slice -- sample 3 of every video,
    rows 768..1023 of the sample-major block, in the middle of the 2048-row launches -- so the gradients are exactly that slice's, and float64
    autograd through oracle/s2vt_torch.py on those 256 rows (their own dropout streams: video j, sample 3) is affordable on the host.
    Compared: embed_word_W (the vocabulary projection, tn128x128 over 71,680 reduction rows), all of lstm2_W (its h rows are the per-step
    recurrence's product), Wemb (the gathered embedding gradient) and the two biases beside them."""
    import torch
    from s2vt_amd import hostglue
    from oracle import s2vt_torch as T
    mdl, video = ref
    s, _ = mdl.sample(video, K, False, seed=35)
    N, k0 = K * B, 3
    rows = slice(k0 * B, (k0 + 1) * B)
    s_h = s.cpu().numpy()
    mask_h = np.zeros((N, TC), np.float32)
    mask_h[rows] = hostglue.masks_from_ids(s_h[rows])
    g = torch.Generator().manual_seed(4)
    r = torch.rand(N, generator=g) * 2; b = (torch.rand(B, generator=g) * 2).repeat(K)
    mdl.global_step = 0
    st = mdl.reinforce_update(video, s, torch.as_tensor(mask_h).cuda(), r.cuda(), b.cuda(), lr=0.0, clip_norm=5.0)
    assert mdl._ctx[8] == TC and mdl._ctx[9] is None                       # all 35 steps unrolled, every row in the vocabulary products
    assert float(st.mask_sum) == float(mask_h.sum())
    names = ("embed_word_W", "embed_word_b", "lstm2_W", "lstm2_b", "Wemb")
    got = {n: mdl.store.g[n].cpu().numpy().astype(np.float64) for n in names}
    # float64 reference on the slice
    p = {n: mdl.store.p[n].cpu().numpy() for n in mdl.store.names}
    pt = T.to_torch(p, torch.float64, False)
    for n in names:
        pt[n].requires_grad_(True)
    keep = mdl.dropout_rate
    dseed = mdl.dropout_seed + 104729 * 0
    drop = oracle.dropout_masks(dseed, np.arange(B, dtype=np.int32), np.full(B, k0, np.int32), keep, H, TV, TC)
    cap = s_h[rows].astype(np.int32)
    lg = T.teacher_forced(pt, video.cpu().double(), cap, drop, keep)
    loss = T.pg_loss(lg, cap, mask_h[rows], r[rows].numpy(), b[rows].numpy())
    assert abs(float(st.loss) - float(loss.detach())) <= 1e-4 * max(1.0, abs(float(loss.detach())))
    loss.backward()
    for n in names:
        rg = pt[n].grad.numpy()
        scale = np.abs(rg).max()
        assert scale > 0, n
        assert np.abs(got[n] - rg).max() <= 3e-4 * scale + 1e-12, (n, float(np.abs(got[n] - rg).max() / scale))
