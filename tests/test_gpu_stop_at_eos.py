"""The opt-in early-exit sampler (s2vt_sample_ex, S2VT_SAMPLE_STOP_AT_EOS): rows that have emitted <eos> leave the decode loop.
Ids up to and including a row's first <eos> must be bit-identical to the reference-faithful sampler (which keeps sampling every row
for all Tc steps, reinforcement_multisampling_tf_s2vt.py:318-337), ids behind it 0; the masks -- and so the REINFORCE update -- are
the same.  The embedding bias of <eos> is raised so that samples end at realistic lengths (a random-initialised model never stops)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims,B,K,eos_bias", [
    (dict(dim_image=64, n_words=131, word_dim=24, lstm_dim=32, n_video_lstm_step=3, n_caption_lstm_step=9), 5, 3, 3.5),
    (dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=12), 16, 2, 4.0),
    (dict(dim_image=1536, n_words=12000, word_dim=500, lstm_dim=1000, n_video_lstm_step=5, n_caption_lstm_step=20), 64, 5, 7.5),
    (dict(dim_image=96, n_words=300, word_dim=20, lstm_dim=48, n_video_lstm_step=2, n_caption_lstm_step=7), 100, 3, 4.5),   # R = 400: several scan chunks
])
def test_stop_at_eos_ids_equal_up_to_first_eos(gpu, dims, B, K, eos_bias):
    import torch
    from s2vt_amd import hostglue, model as M
    mdl = M.Video_Caption_Generator(dims["dim_image"], dims["n_words"], dims["word_dim"], dims["lstm_dim"], B, 0, dims["n_video_lstm_step"],
                                    dims["n_caption_lstm_step"], seed=3, multisample=K)
    mdl.store.p["embed_word_b"][0] = eos_bias                      # P(<eos>) per step ~ 1/7 .. 1/4
    rng = np.random.default_rng(1)
    video = torch.as_tensor(np.abs(rng.standard_normal((B, dims["n_video_lstm_step"], dims["dim_image"])) * 0.5).astype(np.float32)).cuda()
    for seed in (11, 12):
        s_ref, g_ref = mdl.sample(video, K, True, seed=seed)
        s_ref, g_ref = s_ref.cpu().numpy(), g_ref.cpu().numpy()
        s_eos, g_eos = mdl.sample(video, K, True, seed=seed, stop_at_eos=True)
        s_eos, g_eos = s_eos.cpu().numpy(), g_eos.cpu().numpy()
        for ref, got in ((s_ref, s_eos), (g_ref, g_eos)):
            mask = hostglue.masks_from_ids(ref).astype(bool)      # up to and including the first <eos>
            assert mask.sum() < mask.size                         # the test must see rows that stop early ...
            assert np.array_equal(got[mask], ref[mask])           # ... identical where the objective looks
            assert (got[~mask] == 0).all()                        # and <eos> behind it
            assert np.array_equal(hostglue.masks_from_ids(got), hostglue.masks_from_ids(ref))
    assert gpu.chain_timeouts() == 0


def test_stop_at_eos_update_is_the_same_update(gpu):
    """reinforce_update on the early-exit sampler's ids == on the faithful sampler's ids (same mask, same positions)."""
    import torch
    from s2vt_amd import model as M
    B, K = 8, 3
    mk = lambda: M.Video_Caption_Generator(128, 260, 32, 64, B, 0, 5, 10, seed=5, multisample=K, dropout_rate=0.9)
    a, b = mk(), mk()
    for m_ in (a, b):
        m_.store.p["embed_word_b"][0] = 4.0
    rng = np.random.default_rng(2)
    video = torch.as_tensor(np.abs(rng.standard_normal((B, 5, 128)) * 0.5).astype(np.float32)).cuda()
    r = (rng.random(K * B) * 2).astype(np.float32); bl = np.tile((rng.random(B) * 2).astype(np.float32), K)
    sa, _ = a.sample(video, K, True, seed=9)
    a.reinforce_update(video, sa, None, r, bl, lr=1e-2, reuse_sampler_state=True)       # (the sampler workspace is shared: update before the next sample)
    sb, _ = b.sample(video, K, True, seed=9, stop_at_eos=True)
    b.reinforce_update(video, sb, None, r, bl, lr=1e-2, reuse_sampler_state=True)
    assert float((a.store.theta - b.store.theta).abs().max()) <= 1e-6
