"""The reward scorer's parity tests (tests/test_ciderd.py: the C++ CIDEr-D on token ids vs the Python restatement of the
published pyciderevalcap algorithm on the reference's own MSVD sentences; they need no GPU) run again under `-m gpu`, so
that the GPU box's record (GPUTEST) covers SURVEY 8(f)1 and shows libs2vt_host.so loaded there -- and, on the GPU, that
the scorer's rewards are the ones train_rl feeds the update with (ids produced by the sampler, scored in place)."""
import numpy as np
import pytest

import test_ciderd as host_tests

pytestmark = pytest.mark.gpu


def test_ciderd_ids_match_string_restatement_on_the_gpu_box():
    host_tests.test_ciderd_ids_match_string_restatement()


def test_ciderd_exact_reference_scores_high_and_bad_args_on_the_gpu_box():
    host_tests.test_ciderd_exact_reference_scores_high_and_bad_args()


def test_sampler_ids_are_scored_in_place(gpu):
    """ids straight from s2vt_sample (int32, sample-major rows) -> CiderD.score_ids: finite scores, one per caption, and a
    caption that IS one of its video's references scores above the sampled ones."""
    import torch
    from s2vt_amd import hostglue, reward
    from oracle import s2vt_oracle as orc
    vocab = ["<en_unk>", "a", "man", "woman", "is", "playing", "the", "guitar", "dog", "runs"]
    wordtoix, _ = hostglue.preProBuildWordVocab(vocab)
    refs = [["a man is playing the guitar", "the man is playing"], ["a dog runs", "the dog runs"]]
    scorer = reward.CiderD(refs, wordtoix)
    d = orc.Dims(24, len(wordtoix), 8, 16, 3, 6, 0)
    p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 2).items()}
    video = torch.rand(2, 3, 24, device="cuda")
    s, g = gpu.sample(gpu.make_dims(24, len(wordtoix), 8, 16, 3, 6), gpu.make_params(p), video, 3, seed=4)
    r = scorer.score_ids(s.cpu().numpy(), np.tile(np.arange(2, dtype=np.int32), 3))
    b = scorer.score_ids(g.cpu().numpy(), np.arange(2, dtype=np.int32))
    assert r.shape == (6,) and b.shape == (2,) and np.isfinite(r).all() and np.isfinite(b).all()
    exact = np.zeros((1, 6), np.int32)
    words = "a dog runs".split()
    exact[0, :3] = [wordtoix[w] for w in words]
    assert scorer.score_ids(exact, np.asarray([1], np.int32))[0] > max(r.max(), 1e-6)
