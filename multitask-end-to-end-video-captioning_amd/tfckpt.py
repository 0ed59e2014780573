"""TensorFlow checkpoint FILES <-> {variable name: ndarray}: what `optimistic_restore` of the reference reads
(reinforcement_multisampling_tf_s2vt.py:47-61: tf.train.NewCheckpointReader(save_file).get_variable_to_shape_map(), then a
tf.train.Saver restore of the variables whose names and shapes match) and what its savers write (tf_s2vt.py:440
`tf.train.Saver(max_to_keep=100, write_version=1)`, :560; reinforcement_multisampling_tf_s2vt.py:661 default version 2).

TensorFlow is not installable in this image and the reference ships no checkpoint, so both formats are RESTATED from the
published TensorFlow 1.x sources and are pinned only by round trips through this module's own writers (tests/test_tfckpt.py):

* V2 "tensor bundle" (tensorflow/core/util/tensor_bundle): `<prefix>.index` is a table (the LevelDB SSTable layout,
  tensorflow/core/lib/io/table*.cc) whose key "" holds a BundleHeaderProto and whose other keys are variable names holding
  BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; `<prefix>.data-SSSSS-of-NNNNN` hold the raw
  little-endian bytes.
* V1 (tensorflow/core/util/tensor_slice_writer): ONE table file; key "" holds SavedTensorSlices{meta}, every other key holds
  SavedTensorSlices{data: SavedSlice{name, slice, TensorProto}} with the values in the typed repeated fields (float_val,
  int_val, int64_val, double_val) or tensor_content.  Blocks are usually snappy-compressed.

Only what the reference's variables need is implemented: whole (unpartitioned) tensors of float32 / float64 / int32 / int64.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}
_DT_INV = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


# ------------------------------------------------------------------------------------------------ varints / protobuf wire
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _fields(buf):
    """Yield (field number, wire type, value) of a protobuf message; length-delimited values as memoryview slices."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"protobuf wire type {wt} is not used by checkpoint messages")
        yield fn, wt, v


def _ld(fn: int, payload: bytes) -> bytes:
    return _put_varint((fn << 3) | 2) + _put_varint(len(payload)) + payload


def _vi(fn: int, v: int) -> bytes:
    return _put_varint(fn << 3) + _put_varint(v)


def _shape_proto(shape) -> bytes:          # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
    return b"".join(_ld(2, _vi(1, int(d))) for d in shape)


def _parse_shape(buf):
    dims = []
    for fn, _, v in _fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli), masked
def _crc_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_CRC_T = _crc_table()


def _gf2_shift_matrix(nbytes: int):
    """32 columns: the CRC register after `nbytes` zero bytes, for each single-bit start state (crc 'append zeros' operator)."""
    cols = []
    for bit in range(32):
        c = 1 << bit
        for _ in range(nbytes):
            c = (c >> 8) ^ int(_CRC_T[c & 0xFF])
        cols.append(c)
    return cols


def crc32c(data) -> int:
    """CRC-32C of a bytes-like.  Large inputs are cut into 4 KB pieces whose CRCs advance together as numpy vectors (one
    table step per byte position) and are then combined with the 'append 4096 zero bytes' operator."""
    a = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    n = a.size
    L = 4096
    crc = 0xFFFFFFFF
    full = n // L
    if full >= 8:
        m = a[:full * L].reshape(full, L)
        c = np.zeros(full, np.uint32)
        c[0] = 0xFFFFFFFF                                    # only the first piece starts from the initial register
        for i in range(L):
            c = (c >> np.uint32(8)) ^ _CRC_T[(c ^ m[:, i]) & np.uint32(0xFF)]
        cols = _gf2_shift_matrix(L)
        acc = int(c[0])
        for j in range(1, full):
            x, s = acc, 0
            for bit in range(32):
                if x >> bit & 1:
                    s ^= cols[bit]
            acc = s ^ int(c[j])
        crc = acc
        rest = a[full * L:]
    else:
        rest = a
    for b in rest.tolist():
        crc = (crc >> 8) ^ int(_CRC_T[(crc ^ b) & 0xFF])
    return crc ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ snappy (decoder; literal-only encoder)
def snappy_decompress(buf) -> bytes:
    buf = memoryview(buf)
    n, pos = _varint(buf, 0)
    out = bytearray()
    end = len(buf)
    while pos < end:
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                k = ln - 59
                ln = int.from_bytes(buf[pos:pos + k], "little"); pos += k
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little"); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little"); pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:
            for i in range(ln):                               # overlapping copy: byte by byte
                out.append(out[start + i])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


def snappy_compress_literal(data: bytes) -> bytes:
    """A valid snappy stream made of literal elements only (tests: exercises the decoder's framing)."""
    out = bytearray(_put_varint(len(data)))
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + 65536]
        ln = len(chunk) - 1
        if ln < 60:
            out.append(ln << 2)
        else:
            k = (ln.bit_length() + 7) // 8
            out.append((59 + k) << 2)
            out += ln.to_bytes(k, "little")
        out += chunk
        pos += len(chunk)
    return bytes(out)


# ------------------------------------------------------------------------------------------------ the table (SSTable) layout
def _read_block(f, off, size):
    f.seek(off)
    raw = f.read(size + 5)
    body, ctype = raw[:size], raw[size]
    if ctype == 1:
        body = snappy_decompress(body)
    elif ctype != 0:
        raise ValueError(f"unknown block compression {ctype}")
    return body


def _block_entries(body):
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    limit = len(body) - 4 - 4 * nrestart
    pos, key = 0, b""
    mv = memoryview(body)
    while pos < limit:
        shared, pos = _varint(mv, pos)
        non_shared, pos = _varint(mv, pos)
        vlen, pos = _varint(mv, pos)
        key = key[:shared] + bytes(mv[pos:pos + non_shared]); pos += non_shared
        yield key, mv[pos:pos + vlen]
        pos += vlen


def _table_items(path):
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError(f"{path}: too short for a TensorFlow table")
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
            raise ValueError(f"{path}: not a TensorFlow checkpoint table (bad magic)")
        mv = memoryview(footer)
        _, p = _varint(mv, 0); _, p = _varint(mv, p)          # metaindex handle
        ioff, p = _varint(mv, p); isz, p = _varint(mv, p)     # index handle
        for _, handle in list(_block_entries(_read_block(f, ioff, isz))):
            boff, q = _varint(handle, 0); bsz, q = _varint(handle, q)
            for k, v in _block_entries(_read_block(f, boff, bsz)):
                yield k, bytes(v)


class _TableWriter:
    """Blocks of <= ~4 KB of entries, restart interval 1 (no key sharing), index block, empty metaindex, footer."""

    def __init__(self, f, compress=False):
        self.f, self.compress = f, compress
        self.block, self.restarts, self.last_key, self.index = bytearray(), [], b"", []

    def _emit(self, body: bytes):
        ctype = 0
        if self.compress:
            body, ctype = snappy_compress_literal(body), 1
        off = self.f.tell()
        self.f.write(body)
        self.f.write(bytes([ctype]) + struct.pack("<I", _mask(crc32c(body + bytes([ctype])))))
        return off, len(body)

    def _finish_block(self):
        if not self.restarts:
            return
        body = bytes(self.block) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        off, sz = self._emit(body)
        self.index.append((self.last_key, _put_varint(off) + _put_varint(sz)))
        self.block, self.restarts = bytearray(), []

    def add(self, key: bytes, value: bytes):
        assert key >= self.last_key, "table keys must be added in sorted order"
        self.restarts.append(len(self.block))
        self.block += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(value)) + key + value
        self.last_key = key
        if len(self.block) >= 4096:
            self._finish_block()

    def finish(self):
        self._finish_block()
        moff, msz = self._emit(struct.pack("<I", 0) + struct.pack("<I", 1))      # empty metaindex block: one restart at 0
        body = bytearray()
        restarts = []
        for k, h in self.index:
            restarts.append(len(body))
            body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(h)) + k + h
        ibody = bytes(body) + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", max(len(restarts), 1))
        if not restarts:
            ibody = struct.pack("<I", 0) + struct.pack("<I", 1)
        ioff, isz = self._emit(ibody)
        handles = _put_varint(moff) + _put_varint(msz) + _put_varint(ioff) + _put_varint(isz)
        self.f.write(handles + b"\0" * (40 - len(handles)) + struct.pack("<Q", _MAGIC))


# ------------------------------------------------------------------------------------------------ V2: tensor bundle
def _read_v2(prefix):
    entries, nshards = {}, 1
    for k, v in _table_items(prefix + ".index"):
        if k == b"":
            for fn, _, x in _fields(v):
                if fn == 1:
                    nshards = x
            continue
        e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "sliced": False}
        for fn, _, x in _fields(v):
            if fn == 1: e["dtype"] = x
            elif fn == 2: e["shape"] = _parse_shape(x)
            elif fn == 3: e["shard"] = x
            elif fn == 4: e["offset"] = x
            elif fn == 5: e["size"] = x
            elif fn == 7: e["sliced"] = True
        entries[k.decode()] = e
    out = {}
    files = {}
    try:
        for name, e in entries.items():
            if e["sliced"] or e["dtype"] not in _DT:
                continue                                       # partitioned variables / other dtypes: not the reference's
            if e["shard"] not in files:
                files[e["shard"]] = open(f"{prefix}.data-{e['shard']:05d}-of-{nshards:05d}", "rb")
            f = files[e["shard"]]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            out[name] = np.frombuffer(raw, dtype=np.dtype(_DT[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def write_checkpoint_v2(prefix: str, variables: dict):
    """`<prefix>.index` + `<prefix>.data-00000-of-00001` in the tensor-bundle layout (what tf.train.Saver's default writes)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    names = sorted(variables, key=lambda s: s.encode())
    offs = {}
    with open(prefix + ".data-00000-of-00001", "wb") as d:
        for n in names:
            a = np.asarray(variables[n])                          # (np.ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_INV:
                raise TypeError(f"{n}: dtype {a.dtype} is not supported")
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            offs[n] = (d.tell(), len(raw), _mask(crc32c(raw)), a)
            d.write(raw)
    with open(prefix + ".index", "wb") as f:
        t = _TableWriter(f)
        t.add(b"", _vi(1, 1) + _vi(2, 0) + _ld(3, _vi(1, 1)))                      # BundleHeaderProto: num_shards 1, LITTLE, VersionDef{producer 1}
        for n in names:
            off, size, crc, a = offs[n]
            e = _vi(1, _DT_INV[a.dtype]) + _ld(2, _shape_proto(a.shape)) + _vi(3, 0) + _vi(4, off) + _vi(5, size) \
                + _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
            t.add(n.encode(), e)
        t.finish()


# ------------------------------------------------------------------------------------------------ V1: tensor slice writer
def _tensor_proto_values(buf):
    dtype, shape, content = 0, (), None
    vals = {5: [], 6: [], 7: [], 10: []}                       # float_val, double_val, int_val, int64_val
    for fn, wt, v in _fields(buf):
        if fn == 1: dtype = v
        elif fn == 2: shape = _parse_shape(v)
        elif fn == 4: content = bytes(v)
        elif fn in vals:
            if wt == 2:                                         # packed
                if fn == 5: vals[5].append(np.frombuffer(v, "<f4"))
                elif fn == 6: vals[6].append(np.frombuffer(v, "<f8"))
                else:
                    mv, pos, xs = memoryview(v), 0, []
                    while pos < len(mv):
                        x, pos = _varint(mv, pos)
                        xs.append(x if x < (1 << 63) else x - (1 << 64))
                    vals[fn].append(np.asarray(xs, np.int64))
            elif fn == 5: vals[5].append(np.frombuffer(v, "<f4"))
            elif fn == 6: vals[6].append(np.frombuffer(v, "<f8"))
            else: vals[fn].append(np.asarray([v if v < (1 << 63) else v - (1 << 64)], np.int64))
    if dtype not in _DT:
        return None
    if content is not None:
        return np.frombuffer(content, dtype=np.dtype(_DT[dtype]).newbyteorder("<")).reshape(shape).copy()
    src = {1: 5, 2: 6, 3: 7, 9: 10}[dtype]
    flat = np.concatenate(vals[src]) if vals[src] else np.zeros(0)
    return flat.astype(_DT[dtype]).reshape(shape)


def _read_v1(path):
    out = {}
    for k, v in _table_items(path):
        if k == b"":
            continue                                            # SavedTensorSlices{meta}
        for fn, _, x in _fields(v):
            if fn != 2:                                         # SavedTensorSlices.data
                continue
            name, tensor = None, None
            for f2, _, y in _fields(x):                         # SavedSlice {name = 1, slice = 2, data = 3}
                if f2 == 1: name = bytes(y).decode()
                elif f2 == 3: tensor = _tensor_proto_values(y)
            if name is not None and tensor is not None and name not in out:
                out[name] = tensor
    return out


def write_checkpoint_v1(path: str, variables: dict, compress=True):
    """One table file in the tensor-slice-writer layout (what tf.train.Saver(write_version=1) of tf_s2vt.py:440 writes):
    values in the typed repeated fields, whole-tensor slices, blocks snappy-framed when `compress`."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    names = sorted(variables, key=lambda s: s.encode())
    with open(path, "wb") as f:
        t = _TableWriter(f, compress=compress)
        meta = b""
        for n in names:
            a = np.asarray(variables[n])
            ext = b"".join(_ld(1, b"") for _ in a.shape)       # TensorSliceProto: one empty Extent per dimension = the full range
            meta += _ld(1, _ld(1, n.encode()) + _ld(2, _shape_proto(a.shape)) + _vi(3, _DT_INV[a.dtype]) + _ld(4, ext))
        t.add(b"", _ld(1, meta))
        for i, n in enumerate(names):
            a = np.asarray(variables[n])
            dt = _DT_INV[a.dtype]
            if dt == 1: payload = _ld(5, a.astype("<f4").tobytes(order="C"))
            elif dt == 2: payload = _ld(6, a.astype("<f8").tobytes(order="C"))
            else: payload = _ld(7 if dt == 3 else 10, b"".join(_put_varint(int(x)) for x in a.reshape(-1)))
            tp = _vi(1, dt) + _ld(2, _shape_proto(a.shape)) + payload
            ext = b"".join(_ld(1, b"") for _ in a.shape)
            # keys only have to sort after "" and in name order for the reader here; TF's ordered-code key (EncodeTensorNameSlice)
            # starts with a zero byte as well
            t.add(b"\0" + n.encode() + b"\0" + struct.pack(">I", i), _ld(2, _ld(1, n.encode()) + _ld(2, ext) + _ld(3, tp)))
        t.finish()


# ------------------------------------------------------------------------------------------------ entry point
def read_checkpoint(path: str) -> dict:
    """{variable name: ndarray} of a TensorFlow checkpoint given as the reference gives it to the Saver: a V2 prefix
    (`…/model-6` with `model-6.index` beside it) or a V1 file.  Also accepts this repository's own `.npz` dumps."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".index"):
        path = path[:-6]
    if os.path.exists(path + ".index"):
        return _read_v2(path)
    if os.path.exists(path):
        return _read_v1(path)
    raise FileNotFoundError(f"{path}: neither a V2 checkpoint prefix (<path>.index) nor a V1 checkpoint file")
