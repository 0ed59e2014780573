"""Dev helper (needs variants/lib_stamp.so = build_variants.sh stamp:"-DS2VT_STAMP", run with S2VT_LIB=...): where the
main loop of the contraction kernel spends its shader clocks, per wave and chunk, on the step's skinny shapes."""
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
NAMES = ["prologue", "frag0-wait", "half-1(mfma+issue)|ld-issue", "vmcnt-wait", "ld-land", "half-2(mfma+land)", "barrier", "drain|ld-barrier", "epilogue", "land0", "land1", "land2", "land3", "land4+"]


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
    per_wave = [x / waves for x in s[:14]]
    tot = sum(per_wave[:9])
    cpw = chunks / waves
    loop = sum(per_wave[1:7])
    print(f"{name}: {us:.1f} us/launch, {waves // reps} waves, {cpw:.1f} chunks/wave, {tot:.0f} clk/wave (loop {loop / cpw:.0f} clk/chunk)")
    print("   " + "  ".join(f"{n}={v:.0f}" + (f"({v / cpw:.0f}/ch)" if 1 <= i <= 7 or i >= 9 else "") for i, (n, v) in enumerate(zip(NAMES, per_wave))), flush=True)


dev = "cuda"
H, E, V = 1000, 500, 12000
torch.manual_seed(0)
W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
W1 = torch.randn(E + H, 4 * H, device=dev) * 0.03
Wemb = torch.randn(V, E, device=dev) * 0.1
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
cfg = int(os.environ.get("CFG", "-1"))
for M in (384, 320, 64):
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    report(f"LSTM2 sampler form M={M} K=1500", lambda: ops.lstm_cell_fwd(ops.operand(None, k=H), ops.operand(Wemb, rowidx=idx), h, c, W2, b2, M, tile_cfg=cfg))
    report(f"LSTM2 recurrent only M={M} K=1000", lambda: ops.lstm_cell_fwd(ops.operand(None, k=H + E), None, h, c, W2, b2, M, tile_cfg=cfg))
vid = torch.zeros(384, dtype=torch.int32, device=dev); sid = torch.zeros(384, dtype=torch.int32, device=dev)
o2 = torch.randn(384, H, device=dev)
report("PICK M=384", lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1))
sidg = -torch.ones(384, dtype=torch.int32, device=dev)
report("PICK M=384 greedy rows only", lambda: ops.vocab_pick(o2, Wout, bout, vid, sidg, 0, 1))
A = torch.randn(6400, 1000, device=dev)
report("STORE logits 6400x1000x12000", lambda: ops.gemm([ops.operand(A)], Wout, None, M=6400), reps=2)
for M in (64, 128):
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    report(f"PICK M={M}", lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1))
    report(f"PICK M={M} greedy rows only", lambda: ops.vocab_pick(o2, Wout, bout, vid, -torch.ones(M, dtype=torch.int32, device=dev), 0, 1))
for M in [int(x) for x in os.environ.get("PICKM", "64").split(",")]:
    o2s = torch.randn(M, H, device=dev)
    for pc in [int(x) for x in os.environ.get("PICKCFG", "0,6").split(",")]:
        report(f"PICK M={M} cfg {pc}", lambda: ops.vocab_pick(o2s, Wout, bout, vid, sid, 0, 1, tile_cfg=pc))
