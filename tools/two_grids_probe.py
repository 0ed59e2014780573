"""dev probe (variants/lib_noorder.so: chain.hip with the one-persistent-grid-at-a-time ordering switched off by S2VT_NO_CHAIN_ORDER=1): two independent
64-row forward recurrences (LSTM1- and LSTM2-shaped: H = 1000, T = 19) run one after the other on one stream vs side by side on two streams -- what
would a time-chunked pipeline of the two cells of an unroll gain?  Prints per-pair wall times and the arrival-timeout count."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
L = s2vt_amd.lib()
M, H, T = int(os.environ.get("TG_M", "64")), 1000, int(os.environ.get("TG_T", "19"))
dev = "cuda"
g = torch.Generator().manual_seed(0)
def mk():
    W = ((torch.rand(H, 4 * H, generator=g) * 2 - 1) * 0.05).to(dev)
    b = torch.zeros(4 * H, device=dev)
    cin = ((torch.rand(T, M, 4 * H, generator=g) * 2 - 1) * 0.5).to(dev)
    Ch = torch.zeros(T + 1, M, H, device=dev); Hh = torch.zeros(T + 1, M, H, device=dev)
    gates = torch.empty(T, M, 4 * H, device=dev)
    ws = torch.empty(L.s2vt_lstm_recurrence_scratch_bytes(H) // 4 + 64, dtype=torch.float32, device=dev)
    return W, b, cin, Ch, Hh, gates, ws
A, B = mk(), mk()
def launch(x, stream):
    W, b, cin, Ch, Hh, gates, ws = x
    rc = L.s2vt_lstm_recurrence_fwd(W.data_ptr(), 0, b.data_ptr(), cin.data_ptr(), cin.stride(0), cin.stride(1), T, Ch.data_ptr(), Hh.data_ptr(), gates.data_ptr(), None,
                                    M, H, T, 1.0, 0, None, None, 0, 1, ws.data_ptr(), ws.numel() * 4, C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def serial():
    launch(A, s0); launch(B, s0)
def parallel():
    e = torch.cuda.Event(); e.record(s0)
    s1.wait_event(e); s2.wait_event(e)
    launch(A, s1); launch(B, s2)
    e1 = torch.cuda.Event(); e2 = torch.cuda.Event(); e1.record(s1); e2.record(s2)
    s0.wait_event(e1); s0.wait_event(e2)
one = timed(lambda: launch(A, s0))
ser = timed(serial)
refA = A[4].clone(); refB = B[4].clone()
par = timed(parallel)
ok = torch.equal(refA, A[4]) and torch.equal(refB, B[4])
print(f"M={M} T={T}: one chain {one:.1f} us, two in series {ser:.1f} us, two side by side {par:.1f} us (x{par / one:.2f} of one), states identical {ok}, "
      f"timeouts {ops.chain_timeouts()}, order off: {os.environ.get('S2VT_NO_CHAIN_ORDER')}")
