"""Feature-file reader + caption index (CPU): the C++ reader reproduces what the reference's parser yields
(float(str) of every field, grouped per video in file order), the .npy cache round-trips, ragged files are
rejected as the reference's assert does."""
import os

import numpy as np
import pytest


def _write_csv(path, rng, videos=("vid7", "vid12", "vid3"), tv=3, d=10):
    rows = {}
    with open(path, "w") as f:
        for v in videos:
            for k in range(tv):
                x = rng.standard_normal(d).astype(np.float32)
                rows.setdefault(v, []).append(x)
                f.write(f"{v}_frame_{k}," + ",".join(repr(float(t)) for t in x) + "\n")
    return rows


def test_feature_csv_reader_and_cache(tmp_path):
    import s2vt_amd
    from s2vt_amd import data
    rng = np.random.default_rng(0)
    p = str(tmp_path / "feat.txt")
    rows = _write_csv(p, rng)
    st = data.FeatureStore.from_csv(p)
    assert st.video_ids == ["vid7", "vid12", "vid3"] and st.features.shape == (3, 3, 10)
    for v in rows:
        assert np.array_equal(st[v], np.stack(rows[v]))                  # exact: repr(float32) round-trips
    st2 = data.FeatureStore.from_csv(p)                                   # served from the .npy cache
    assert os.path.exists(p + ".f32.npy") and np.array_equal(st2.features, st.features)
    b = st.batch(["vid3", "vid7"], pinned=False)
    assert np.array_equal(b[0], st["vid3"]) and np.array_equal(b[1], st["vid7"])


def test_ragged_feature_file_rejected(tmp_path):
    import s2vt_amd
    from s2vt_amd import data
    p = str(tmp_path / "bad.txt")
    open(p, "w").write("vid1_frame_0,1.0,2.0\nvid1_frame_1,1.0\n")
    with pytest.raises(IOError):
        data.FeatureStore.from_csv(p, cache=False)
    p2 = str(tmp_path / "uneven.txt")
    open(p2, "w").write("vid1_frame_0,1.0,2.0\nvid1_frame_1,1.0,3.0\nvid2_frame_0,1.0,2.0\n")
    with pytest.raises(AssertionError):
        data.FeatureStore.from_csv(p2, cache=False)


def test_caption_index(tmp_path):
    import s2vt_amd
    from s2vt_amd import data
    p = str(tmp_path / "sents.txt")
    open(p, "w").write("vid1\ta man is cooking\nvid2\ta cat\nvid1\tsomeone cooks food\n")
    sents = data.read_sentences(p)
    assert sents.shape == (3, 2) and sents[2, 0] == "vid1"
    ix = data.CaptionIndex(sents)
    assert ix.get_captions("vid1") == ["a man is cooking", "someone cooks food"] and ix.video_ids == ["vid1", "vid2"]
