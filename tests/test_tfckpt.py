"""tfckpt.py: TensorFlow checkpoint files (V2 tensor bundle, V1 tensor-slice table) <-> {name: ndarray}.  TensorFlow is not
installable here and the reference ships no checkpoint: the formats are restated from the published sources and pinned by
known-answer tests of their pieces (CRC-32C check value, snappy framing incl. overlapping copies, the table layout) and by
round trips through the module's own writers.  Then optimistic_restore from such FILES into a ParamStore."""
import os
import struct

import numpy as np
import pytest

import s2vt_amd
from s2vt_amd import tfckpt


def test_crc32c_known_answers_and_vector_path():
    assert tfckpt.crc32c(b"123456789") == 0xE3069283              # the CRC-32C (Castagnoli) check value
    assert tfckpt.crc32c(b"") == 0
    assert tfckpt.crc32c(b"\0" * 32) == 0x8A9136AA                 # RFC 3720 B.4: 32 bytes of zeros
    assert tfckpt.crc32c(bytes(range(32))) == 0x46DD794E           # RFC 3720 B.4: 00..1f
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, 9 * 4096 + 123, dtype=np.uint8).tobytes()          # takes the pieces-in-parallel path

    def slow(b):
        c = 0xFFFFFFFF
        for x in b:
            c = (c >> 8) ^ int(tfckpt._CRC_T[(c ^ x) & 0xFF])
        return c ^ 0xFFFFFFFF
    assert tfckpt.crc32c(big) == slow(big)
    assert tfckpt._mask(0) == 0xA282EAD8


def test_snappy_decoder_literals_and_copies():
    data = b"s2vt " * 3000
    assert tfckpt.snappy_decompress(tfckpt.snappy_compress_literal(data)) == data
    # hand-made stream: literal "abc", then a 1-byte-offset copy (len 7, offset 3: overlapping), then a 2-byte-offset copy
    stream = bytes([13]) + bytes([2 << 2]) + b"abc" + bytes([((7 - 4) << 2) | 1, 3]) + bytes([((3 - 1) << 2) | 2]) + struct.pack("<H", 10)
    assert tfckpt.snappy_decompress(stream) == b"abcabcabcaabc"
    with pytest.raises(ValueError):
        tfckpt.snappy_decompress(bytes([5, 0 << 2]) + b"a")        # announces 5 bytes, delivers 1


def _variables(rng):
    return {
        "Wemb": rng.standard_normal((37, 12)).astype(np.float32),
        "s2vt/LSTM1/basic_lstm_cell/weights": rng.standard_normal((20, 32)).astype(np.float32),
        "s2vt/LSTM1/basic_lstm_cell/weights/Adam": rng.standard_normal((20, 32)).astype(np.float32),
        "embed_word_b": rng.standard_normal(37).astype(np.float32),
        "beta1_power": np.float32(0.9 ** 8),
        "g_step": np.int64(7),
        "big": rng.standard_normal((300, 70)).astype(np.float32),  # > one 4 KB table block of index entries is not needed; > 64 KB of data is
        "counts": np.arange(-5, 40, dtype=np.int32).reshape(5, 9),
    }


def test_v2_bundle_round_trip_and_layout(tmp_path):
    v = _variables(np.random.default_rng(1))
    prefix = str(tmp_path / "reinforce_multisample_model-6")
    tfckpt.write_checkpoint_v2(prefix, v)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    with open(prefix + ".index", "rb") as f:
        assert struct.unpack("<Q", f.read()[-8:])[0] == 0xDB4775248B80FB57         # the table magic
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(np.asarray(a).nbytes for a in v.values())
    for path in (prefix, prefix + ".index"):
        got = tfckpt.read_checkpoint(path)
        assert set(got) == set(v)
        for k in v:
            assert got[k].dtype == np.asarray(v[k]).dtype and got[k].shape == np.shape(v[k]) and np.array_equal(got[k], v[k]), k


@pytest.mark.parametrize("compress", [True, False])
def test_v1_table_round_trip(tmp_path, compress):
    v = _variables(np.random.default_rng(2))
    path = str(tmp_path / "batch_size64_s2vt_model-10")
    tfckpt.write_checkpoint_v1(path, v, compress=compress)
    got = tfckpt.read_checkpoint(path)
    assert set(got) == set(v)
    for k in v:
        assert got[k].dtype == np.asarray(v[k]).dtype and got[k].shape == np.shape(v[k]) and np.array_equal(got[k], v[k]), k


def test_optimistic_restore_from_tensorflow_files(tmp_path):
    """reinforcement_multisampling_tf_s2vt.py:667: optimistic_restore(sess, '<dir>/batch_size64..._model-10') -- a checkpoint
    FILE, matched by name and shape.  An XE checkpoint restored into a REINFORCE graph brings the model variables only (the
    reference's XE saver, tf_s2vt.py:440, is created before the optimizer and holds no slots: Adam starts fresh); a foreign
    counter name does not match; optimizer_state=True takes the slots anyway."""
    import torch
    from s2vt_amd import model as M, train_common as tc

    class Stub:                                                   # what optimistic_restore needs of a model, on the CPU
        def __init__(self, store):
            self.store, self.global_step, self.adam_t = store, 0, 0

        def set_step(self, g, a=None):
            self.global_step, self.adam_t = int(g), int(g if a is None else a)
    shapes = M.param_shapes(8, 11, 4, 4)
    a = M.ParamStore(shapes, torch.device("cpu"))
    M.init_reference(a, seed=3)
    a.m.uniform_(-1, 1); a.v.uniform_(0, 1)
    sd = a.state_dict(global_step=12, adam_t=12, step_name="Variable")
    sd.pop("global_step")
    v2 = str(tmp_path / "xe_model-3"); v1 = str(tmp_path / "xe_model_v1-3")
    tfckpt.write_checkpoint_v2(v2, sd)
    tfckpt.write_checkpoint_v1(v1, sd)
    for path in (v2, v1):
        b = Stub(M.ParamStore(shapes, torch.device("cpu")))
        loaded = tc.optimistic_restore(b, path, step_names=("g_step",))           # a REINFORCE graph: its counter is 'g_step'
        assert "s2vt/LSTM2/basic_lstm_cell/weights" in loaded and "s2vt/LSTM2/basic_lstm_cell/weights/Adam_1" not in loaded and "Variable" not in loaded
        assert b.global_step == 0 and b.adam_t == 0 and float(b.store.m.abs().max()) == 0.0
        for n in a.names:
            assert torch.equal(a.p[n], b.store.p[n])
        b2 = Stub(M.ParamStore(shapes, torch.device("cpu")))
        loaded = tc.optimistic_restore(b2, path, step_names=("g_step",), optimizer_state=True)
        assert "s2vt/LSTM2/basic_lstm_cell/weights/Adam_1" in loaded and b2.global_step == 0 and b2.adam_t == 12
        for n in a.names:
            assert torch.equal(a._view(a.m, n), b2.store._view(b2.store.m, n))
        c = Stub(M.ParamStore(shapes, torch.device("cpu")))
        tc.optimistic_restore(c, path)                                            # an XE graph resuming: counter restored too
        assert c.global_step == 12 and c.adam_t == 12
    with pytest.raises(FileNotFoundError):
        tfckpt.read_checkpoint(str(tmp_path / "nothing-here"))
