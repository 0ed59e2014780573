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


def test_frame_ticks_and_image_pipeline(tmp_path):
    """Frame sampling rule and image preprocessing of the end-to-end scripts (e2e_tf_s2vt.py:388-398,436-447)."""
    from PIL import Image
    import s2vt_amd
    from s2vt_amd import data
    assert data.frame_ticks(30, 5) == [1, 8, 15, 22, 29]                 # step (30-2)//4 = 7
    assert data.frame_ticks(6, 5) == [1, 2, 3, 4, 5]
    assert data.frame_ticks(5, 5) == [1] * 5                             # too short: step 0 -> frame 1 repeated
    assert data.frame_ticks(11, 5) == [1, 3, 5, 7, 9]
    rng = np.random.default_rng(0)
    sent = tmp_path / "s.txt"
    with open(sent, "w") as f:
        for v, n in (("vidA", 12), ("vidB", 7)):
            os.makedirs(tmp_path / "frames" / v)
            for k in range(1, n + 1):
                Image.fromarray(rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)).save(tmp_path / "frames" / v / f"{k:06d}.jpg")
            f.write(f"{v}\ta man is walking\n{v}\ta person walks\n")
    sents, frames = data.get_video_frame_caption_pair(str(sent), str(tmp_path / "frames"), 3)
    assert sents.shape == (4, 2) and list(frames) == ["vidA", "vidB"]
    assert [os.path.basename(p) for p in frames["vidA"]] == ["000001.jpg", "000006.jpg", "000011.jpg"]
    assert [os.path.basename(p) for p in frames["vidB"]] == ["000001.jpg", "000003.jpg", "000005.jpg"]
    x = data.image_reading_processing([frames["vidA"], frames["vidB"]], width=16, height=12)
    assert x.shape == (2, 3, 3, 12, 16) and x.dtype == np.float32 and -1.0 <= x.min() and x.max() <= 1.0
    # a solid-colour image survives decode + resize: 2 * (v / 255) - 1 per channel, RGB order
    Image.new("RGB", (40, 40), (255, 0, 128)).save(tmp_path / "solid.png")
    y = data.image_reading_processing([[str(tmp_path / "solid.png")]], 8, 8)[0, 0]
    assert np.allclose(y[0], 1.0) and np.allclose(y[1], -1.0) and np.allclose(y[2], 2 * 128 / 255 - 1, atol=1e-6)


def test_frame_ticks_match_the_reference_function(tmp_path):
    """Which frames of a video feed the CNN (e2e_tf_s2vt.py:376-412), pinned to the reference's own get_video_feature_caption_pair executed in the
    build container on a scratch tree (tools/make_fixtures.py -> tests/golden/hostglue.json): 17 video lengths x 3 values of num_frame_per_video,
    incl. videos too short for the step (frame 1 repeated) and a single-frame video; and the same through the file-system entry point."""
    import json
    from s2vt_amd import data
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostglue.json")))
    cases = G["frame_tick_cases"]
    assert len(cases) == 3
    for c in cases:
        n = c["num_frame_per_video"]
        for cnt, ticks in c["ticks"].items():
            assert data.frame_ticks(int(cnt), n) == ticks, (n, cnt)
    c = cases[0]
    sent = tmp_path / "sents.txt"
    with open(sent, "w") as f:
        for cnt in (3, 7, 11, 30):
            d = tmp_path / "frames" / f"vid{cnt}"
            d.mkdir(parents=True)
            for k in range(1, cnt + 1):
                (d / f"{k:06d}.jpg").touch()
            f.write(f"vid{cnt}\ta caption of video {cnt}\n")
    sents, frames = data.get_video_frame_caption_pair(str(sent), str(tmp_path / "frames"), c["num_frame_per_video"])
    assert sents.shape == (4, 2) and sents[0].tolist() == ["vid3", "a caption of video 3"]
    for cnt in (3, 7, 11, 30):
        assert [int(os.path.basename(p)[:6]) for p in frames[f"vid{cnt}"]] == c["ticks"][str(cnt)]


def test_feature_file_parser_matches_the_reference_function(tmp_path):
    """The feature text file (writer tf_feature_extract.py:153-154) as the reference's own parser reads it (tf_s2vt.py:324-345, executed in the build
    container): video id = the text before the first '_', frames in FILE order (interleaved videos, frame numbers out of order), and the values
    bit for bit what feeding the parsed strings to a float32 placeholder gives -- incl. 0, a denormal, FLT_MAX, scientific notation, 17 digits."""
    import json
    from s2vt_amd import data
    c = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostglue.json")))["feature_file_case"]
    f = tmp_path / "feats.txt"
    f.write_text(c["file_text"])
    fs = data.FeatureStore.from_csv(str(f), cache=False)
    assert list(fs.video_ids) == c["video_order"]
    for i, v in enumerate(c["video_order"]):
        want = np.asarray(c["features_f32_bits"][v], np.uint32)
        got = np.ascontiguousarray(fs.features[i]).view(np.uint32)
        assert got.shape == want.shape and np.array_equal(got, want), v
