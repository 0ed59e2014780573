"""Self-critical reward on token ids: ctypes binding of libs2vt_host.so (include/s2vt_host.h).

Stands where the reference calls evaluate_captions_cider(ref_decoded, decoded) (cider_evaluation.py:60-87,
driven at reinforcement_multisampling_tf_s2vt.py:784-803): r for the K*B sampled captions, b for the B
greedy captions, both against ALL ground-truth captions of the row's video.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def host_lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("S2VT_HOST_LIB") or os.path.join(_HERE, "libs2vt_host.so")   # S2VT_HOST_LIB: the sanitizer build (tests)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build()")
        L = C.CDLL(path)
        L.s2vt_cider_create.restype = C.c_void_p
        L.s2vt_cider_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.s2vt_cider_destroy.restype = None
        L.s2vt_cider_destroy.argtypes = [C.c_void_p]
        L.s2vt_cider_score.restype = C.c_int
        L.s2vt_cider_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.s2vt_cider_num_videos.restype = C.c_int32
        L.s2vt_cider_num_videos.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def tokenize_refs(refs_by_video, wordtoix):
    """Reference strings -> (tokens int32, offsets int64, video_of_ref int32).  Words outside the model
    vocabulary get private ids >= len(wordtoix): they take part in n-grams and document frequencies exactly
    as their strings would, and can never equal a generated token."""
    oov = {}
    base = max(wordtoix.values()) + 1
    toks, offs, vids = [], [0], []
    for v, refs in enumerate(refs_by_video):
        for s in refs:
            for w in s.split():
                i = wordtoix.get(w)
                if i is None:
                    i = oov.setdefault(w, base + len(oov))
                toks.append(i)
            offs.append(len(toks))
            vids.append(v)
    return np.asarray(toks, np.int32), np.asarray(offs, np.int64), np.asarray(vids, np.int32)


class CiderD:
    """CIDEr-D scorer over a fixed reference corpus (one document per video)."""

    def __init__(self, refs_by_video, wordtoix, n_threads: int = 0):
        toks, offs, vids = tokenize_refs(refs_by_video, wordtoix)
        self.n_threads = n_threads
        self._h = host_lib().s2vt_cider_create(toks.ctypes.data, offs.ctypes.data, vids.ctypes.data, len(vids), len(refs_by_video))
        if not self._h:
            raise ValueError("s2vt_cider_create failed (empty corpus or bad indices)")

    def score_ids(self, ids, video_of_row, eos_id: int = 0):
        """ids [N, Tc] int32 (numpy or CPU tensor) -> float32 [N] rewards."""
        ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
        vr = np.ascontiguousarray(np.asarray(video_of_row), dtype=np.int32)
        out = np.empty(ids.shape[0], np.float32)
        rc = host_lib().s2vt_cider_score(self._h, ids.ctypes.data, ids.shape[0], ids.shape[1], eos_id, vr.ctypes.data,
                                         out.ctypes.data, self.n_threads)
        if rc != 0:
            raise ValueError("s2vt_cider_score: bad arguments (video index out of range?)")
        return out

    def close(self):
        if self._h:
            host_lib().s2vt_cider_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
