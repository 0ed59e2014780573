"""Beam-search generator on the device steps: beam 1 is the greedy caption; a wider beam returns a caption whose
reported log-probability is the model's own log-probability of that sentence."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sentence_logprob(mdl, video, sent):
    """log p(sentence | video) from the teacher-forced path (no dropout)."""
    import torch
    from s2vt_amd import ops
    Tc = mdl.n_caption_lstm_step
    cap = np.zeros((1, Tc), np.int32); cap[0, :len(sent)] = sent
    c = torch.as_tensor(cap).cuda()
    vid = torch.zeros(1, dtype=torch.int32, device="cuda")
    logits, _ = ops.teacher_forced_fwd(mdl.dims, mdl.store.params, video, c, 1, 1.0, 0, vid, vid)
    _, lp = ops.softmax_nll_fwd_bwd(logits, c.t().contiguous().view(-1), torch.zeros(Tc, device="cuda"), 0.0)
    return float(lp[:len(sent)].sum())


def test_beam_search_generator(gpu):
    import torch
    from s2vt_amd import model as M
    from s2vt_amd.beam_generator import BeamSearchGenerator
    torch.manual_seed(0)
    mdl = M.Video_Caption_Generator(24, 60, 12, 20, 1, 0, 3, 7, seed=3)
    mdl.store.p["embed_word_b"][0] += 1.0                      # make <eos> reachable so beams finish
    rng = np.random.default_rng(1)
    for trial in range(4):
        video = torch.as_tensor(np.abs(rng.standard_normal((1, 3, 24))).astype(np.float32)).cuda()
        _, g = mdl.sample(video, 0, True)
        greedy = g.cpu().numpy()[0].tolist()
        s1, lp1, _ = BeamSearchGenerator(mdl, 1).generate(video)
        assert s1 == greedy[:len(s1)]                                           # beam 1 == greedy (until it stops at <eos>)
        s3, lp3, sc3 = BeamSearchGenerator(mdl, 3, 0.0).generate(video)
        assert abs(lp3 - _sentence_logprob(mdl, video, s3)) < 1e-3             # the bookkeeping tracks the model's log-prob
        assert s3[-1] == 0 or len(s3) == 7                                      # finished, or ran to n_caption_lstm_step
        assert 0 not in s3[1:-1] and abs(sc3 - lp3) < 1e-6      # only a FIRST-step <eos> is expanded (reference quirk, :262-266); factor 0: score = logprob
    video_ph, sentence, _ = mdl.build_generator(beam_size=3, length_normalization_factor=0.5)
    words = M.Session(mdl).run(sentence, {video_ph: video.cpu().numpy()})
    assert len(words) == 7


def test_generator_unshifted_softmax_quirk(gpu, oracle):
    """SURVEY A9 / tf_s2vt.py:208-209: build_generator picks argmax(exp(l) / sum(exp(l))) with no max shift.  The kernel
    equals the oracle's restatement bit for bit (probabilities and ids); behind the switch the generator returns <eos>
    once a logit overflows exp, where the default (argmax of the logits) keeps the overflowing word."""
    import torch
    from s2vt_amd import model as M, ops
    rng = np.random.default_rng(4)
    for V in (300, 12000, 97):
        l = (rng.standard_normal((6, V)) * 3).astype(np.float32)
        l[0, 7] = 95.0                      # overflow: inf / inf = NaN there, 0 elsewhere -> index 0
        l[1, 7] = 87.5                      # large but finite: still index 7
        l[2, :] = 0.0; l[2, [40, 41, 90]] = 3.0          # exact ties -> lowest index
        l[3, :] = -200.0                    # every exp underflows: 0 / 0 = NaN everywhere -> index 0
        l[4, 5] = 88.9; l[4, 6] = 90.0      # two NaNs
        ref_ids, ref_p = oracle.softmax_unshifted_argmax(l, True)
        ids, p = ops.softmax_unshifted_argmax(torch.as_tensor(l).cuda(), want_probs=True)
        assert ref_ids.tolist()[:5] == [0, 7, 40, 0, 0]
        assert np.array_equal(ids.cpu().numpy(), ref_ids)
        assert np.array_equal(p.cpu().numpy().view(np.uint32), ref_p.view(np.uint32))     # NaNs included
        assert ref_ids[5] == int(np.argmax(l[5]))                                     # ordinary row: the plain argmax
    mdl = M.Video_Caption_Generator(24, 60, 12, 20, 1, 0, 3, 7, seed=3)
    video = np.abs(rng.standard_normal((1, 3, 24))).astype(np.float32)
    sess = M.Session(mdl)
    vp, sent, _ = mdl.build_generator()
    vq, sent_q, _ = mdl.build_generator(unshifted_softmax=True)
    plain = sess.run(sent, {vp: video})
    assert sess.run(sent_q, {vq: video}) == plain                                     # finite logits: same caption
    _, g = mdl.sample(video, 0, True)
    assert plain == g.cpu().numpy()[0].tolist()
    mdl.store.p["embed_word_b"][9] = 120.0                                            # word 9's logit overflows exp at every step
    assert sess.run(sent, {vp: video}) == [9] * 7
    assert sess.run(sent_q, {vq: video}) == [0] * 7
