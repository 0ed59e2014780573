"""Dev helper (variants/lib_stamp.so, S2VT_LIB=...): the LDS-DMA ring tiles against the register-staged ones, per-segment shader clocks.
Loader and MFMA waves both add to the sums: loader segments are ld-issue / vmcnt-wait / ld-barrier, MFMA segments frag0-wait / half-2 (= the chunk's
MFMAs) / barrier."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops, _lib

lib = _lib.lib()
lib.s2vt_stamp_read.argtypes = [C.POINTER(C.c_ulonglong)]
lib.s2vt_stamp_read.restype = C.c_int
NAMES = ["prologue", "frag0-wait", "ld-issue", "vmcnt-wait", "ld-land", "mfma", "barrier", "ld-barrier", "epilogue", "e9", "e10", "e11", "e12", "e13"]


def read():
    buf = (C.c_ulonglong * 16)()
    assert lib.s2vt_stamp_read(buf) == 0
    return list(buf)


def report(name, fn, reps=5):
    for _ in range(2):
        fn()
    read()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    s = read()
    waves, chunks = s[14], s[15]
    cpw = chunks / max(waves, 1)
    print(f"{name}: {us:.1f} us/launch, {waves // reps} waves, {cpw:.1f} chunks/wave")
    print("   sums over ALL waves / (waves * chunks): " + "  ".join(f"{n}={v / max(chunks, 1):.0f}" for n, v in zip(NAMES, s[:14]) if v), flush=True)


dev = "cuda"
H, E, V = 1000, 500, 12000
torch.manual_seed(0)
W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
Wemb = torch.randn(V, E, device=dev) * 0.1
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
for M, cfgs in ((384, (11, 12)), (64, (9, 13, 14))):
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    for cfg in cfgs:
        report(f"LSTM2 sampler form M={M} K=1500 cfg {cfg}", lambda: ops.lstm_cell_fwd(ops.operand(None, k=H), ops.operand(Wemb, rowidx=idx), h, c, W2, b2, M, tile_cfg=cfg))
for M, cfgs in ((384, (4, 7, 8)), (64, (6, 9, 10))):
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    for pc in cfgs:
        report(f"PICK M={M} cfg {pc}", lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=pc))
