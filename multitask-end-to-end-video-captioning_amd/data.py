"""Data side of the callers: feature files, sentence files, vocabulary, per-video reference index.

Mirrors get_video_feature_caption_pair (tf_s2vt.py:324-345, same in reinforcement_multisampling_tf_s2vt.py
:524-545) and the vocabulary read in train() (:605-610), re-designed for the device path:

* the feature TEXT file ("vid<N>_frame_<k>,f0,...") is read ONCE by the C++ reader of libs2vt_host.so into a
  contiguous float32 [n_videos, Tv, d] array and cached next to the file as .npy (+ the video order); the
  reference re-parses it into Python lists of strings on every run;
* batches are gathered into a pinned host buffer so the [B, Tv, d] block goes to HBM in one coalesced copy;
* `CaptionIndex` answers "all captions of video v" from a dict built once; the reference scans all ~48k
  (id, sentence) pairs per video per step (get_captions, reinforcement_multisampling_tf_s2vt.py:600-601).
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import reward


def read_sentences(sent_file):
    """[(video_id, sentence)] as an [n, 2] array of str, file order (tf_s2vt.py:327-331)."""
    sents = []
    with open(sent_file, "r") as f:
        for line in f:
            id_sent = line.strip().split("\t")
            sents.append((id_sent[0], id_sent[1] if len(id_sent) > 1 else ""))
    return np.array(sents, dtype=object).reshape(-1, 2)


def read_vocabulary(vocabulary_file):
    with open(vocabulary_file, "r") as f:
        return [line.rstrip() for line in f]


class FeatureStore:
    """All videos of one feature file as float32 [n_videos, Tv, d]; `store[vid]` / `store.batch(vids)`."""

    def __init__(self, features: np.ndarray, video_ids):
        assert features.ndim == 3 and features.dtype == np.float32 and len(video_ids) == features.shape[0]
        self.features = np.ascontiguousarray(features)
        self.video_ids = list(video_ids)
        self.index = {v: i for i, v in enumerate(self.video_ids)}
        self._pinned = None

    @classmethod
    def from_csv(cls, feature_file, cache=True):
        npy, meta = feature_file + ".f32.npy", feature_file + ".ids.json"
        if cache and os.path.exists(npy) and os.path.exists(meta) and os.path.getmtime(npy) >= os.path.getmtime(feature_file):
            return cls(np.load(npy), json.load(open(meta)))
        L = reward.host_lib()
        L.s2vt_feature_csv_scan.restype = C.c_int
        L.s2vt_feature_csv_scan.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.s2vt_feature_csv_read.restype = C.c_int
        L.s2vt_feature_csv_read.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        n, d = C.c_int64(), C.c_int32()
        rc = L.s2vt_feature_csv_scan(feature_file.encode(), C.byref(n), C.byref(d))
        if rc != 0:
            raise IOError(f"cannot read {feature_file} (code {rc}: -2 open failed, -3 ragged rows)")
        rows = np.empty((n.value, d.value), np.float32)
        ids = np.zeros((n.value, 64), np.uint8)
        rc = L.s2vt_feature_csv_read(feature_file.encode(), n.value, d.value, rows.ctypes.data, ids.ctypes.data, 64)
        if rc != 0:
            raise IOError(f"malformed feature file {feature_file} (code {rc})")
        frame_ids = [bytes(r).split(b"\0", 1)[0].decode() for r in ids]
        order, per = [], {}
        for i, fid in enumerate(frame_ids):                    # video id = text before the first '_' (tf_s2vt.py:336)
            v = fid.split("_")[0]
            if v not in per:
                per[v] = []
                order.append(v)
            per[v].append(i)
        lengths = {len(v) for v in per.values()}
        assert len(lengths) == 1, f"videos with different numbers of frames: {sorted(lengths)}"   # tf_s2vt.py:342
        feats = np.stack([rows[per[v]] for v in order])
        if cache:
            try:
                np.save(npy, feats)
                json.dump(order, open(meta, "w"))
            except OSError:
                pass
        return cls(feats, order)

    def __len__(self):
        return len(self.video_ids)

    def __contains__(self, vid):
        return vid in self.index

    def __getitem__(self, vid):
        return self.features[self.index[vid]]

    def batch(self, vids, pinned=True):
        """[B, Tv, d] float32; a pinned torch tensor when torch + a GPU are present (else numpy)."""
        rows = np.fromiter((self.index[v] for v in vids), np.int64, len(vids))
        if pinned:
            try:
                import torch
                if torch.cuda.is_available():
                    # two staging buffers, used in turn: the training loops stage batch i + 1 while the GPU still works on
                    # batch i, whose asynchronous copy from ITS buffer is known to have run only once step i is synchronised
                    B = len(rows)
                    if self._pinned is None:
                        self._pinned, self._pin_turn = [None, None], 0
                    k = self._pin_turn
                    self._pin_turn ^= 1
                    if self._pinned[k] is None or self._pinned[k].shape[0] < B:
                        self._pinned[k] = torch.empty((B,) + self.features.shape[1:], dtype=torch.float32).pin_memory()
                    out = self._pinned[k][:B]
                    np.take(self.features, rows, axis=0, out=out.numpy())
                    return out
            except ImportError:
                pass
        return self.features[rows]


def get_video_feature_caption_pair(sent_file, feature_file):
    """(sents [n,2], FeatureStore) -- the reference returns (np.array(sents), dict video -> list of Tv rows)."""
    return read_sentences(sent_file), FeatureStore.from_csv(feature_file)


class CaptionIndex:
    """video id -> all of its captions (what get_captions(train_captions, vid) scans for)."""

    def __init__(self, sents):
        self.by_video = {}
        for vid, s in sents:
            self.by_video.setdefault(vid, []).append(s)
        self.video_ids = list(self.by_video)
        self.row = {v: i for i, v in enumerate(self.video_ids)}

    def get_captions(self, vid):
        return self.by_video[vid]

    def refs_by_video(self):
        return [self.by_video[v] for v in self.video_ids]


# ------------------------------------------------------------------------------------------------
# frame side of the end-to-end scripts (e2e_tf_s2vt.py:376-412,436-447)
# ------------------------------------------------------------------------------------------------
def frame_ticks(frame_cnt: int, num_frame_per_video: int):
    """Which frame numbers (1-based file names %06d.jpg) a video contributes: evenly spaced from frame 1 with
    step (frame_cnt - 2) // (n - 1); a video too short for that repeats frame 1 (e2e_tf_s2vt.py:391-395)."""
    step = (frame_cnt - 2) // (num_frame_per_video - 1) if num_frame_per_video > 1 else 0
    if step > 0:
        return list(range(1, min(2 + step * (num_frame_per_video - 1), frame_cnt), step))
    return [1] * num_frame_per_video


def get_video_frame_caption_pair(sent_file, frame_path, num_frame_per_video, prefix=""):
    """(sents [n,2], {video: [frame file, ...]}) -- get_video_feature_caption_pair of the e2e scripts; every video
    must yield the same number of frames (the reference asserts it)."""
    import glob
    import os
    sents = read_sentences(sent_file)
    frames = {}
    for vid in dict.fromkeys(sents[:, 0].tolist()):
        vdir = os.path.join(frame_path, vid)
        cnt = len(glob.glob(os.path.join(vdir, prefix + "*")))
        frames[vid] = [os.path.join(vdir, f"{prefix}{t:06d}.jpg") for t in frame_ticks(cnt, num_frame_per_video)]
    lengths = {len(v) for v in frames.values()}
    if len(lengths) > 1:
        raise ValueError(f"videos yield different frame counts: {sorted(lengths)}")
    return sents, frames


def image_reading_processing(paths, width=299, height=299, out=None):
    """paths: [B][Tv] image files -> float32 [B, Tv, 3, height, width] in [-1, 1] (RGB, bicubic resize,
    2 * (x / 255) - 1: e2e_tf_s2vt.py:436-447; channel-first for the torch CNN).  Decoding uses Pillow -- cv2, which the
    reference uses, is not in this image; the two bicubic kernels differ in the last bits of a pixel."""
    from PIL import Image
    B, Tv = len(paths), len(paths[0])
    if out is None:
        out = np.empty((B, Tv, 3, height, width), np.float32)
    for i, row in enumerate(paths):
        for j, f in enumerate(row):
            with Image.open(f) as im:
                a = np.asarray(im.convert("RGB").resize((width, height), Image.BICUBIC), np.float32)
            out[i, j] = (2.0 * (a / 255.0) - 1.0).transpose(2, 0, 1)
    return out
