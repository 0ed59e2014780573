"""Dev check: the attention forward through the persistent recurrence (attn_chain.hip) vs per-step launches (ops.chain_hold):
every saved activation of the workspace, region by region, bitwise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import attention as A, ops

D, V, H, Tv, Tc, B = 1536, 12000, 1000, int(os.environ.get("TV", "5")), 20, 64
m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, 0.9)
rng = np.random.default_rng(0)
video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
cap = torch.as_tensor(rng.integers(0, V, (B, Tc)).astype(np.int32)).cuda()
vid, sid = m._row_ids(B)


def run(hold):
    ws = ops.attn_workspace(m.dims, B, video.device)
    ws.zero_()
    if hold:
        with ops.chain_hold():
            lg, al, _ = ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, cap, 0.9, 77, vid, sid, want_alphas=True)
    else:
        lg, al, _ = ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, cap, 0.9, 77, vid, sid, want_alphas=True)
    torch.cuda.synchronize()
    return lg.clone(), al.clone(), ws.clone()


def carve():
    off = 0
    out = []

    def take(name, n, sz=4):
        nonlocal off
        b = (n * sz + 255) & ~255
        out.append((name, off, n * sz))
        off += b
    b = B
    take("encidx", Tv * b); take("prev", Tc * b); take("tgt", Tc * b); take("vid", b); take("sid", b)
    take("Vt", Tv * b * H); take("P", Tv * b * H)
    take("hWa", Tc * b * H); take("alpha", Tc * Tv * b); take("asum", Tc * b); take("ctx", Tc * b * H)
    take("G3", Tc * b * 4 * H); take("C3", (Tc + 1) * b * H); take("H3", (Tc + 1) * b * H); take("O3", (Tc + 1) * b * H)
    take("Y", Tc * b * H)
    return out


a = run(False); h = run(True)
print("logits equal", torch.equal(a[0], h[0]), "alphas equal", torch.equal(a[1], h[1]))
for name, off, nb in carve():
    x = a[2][off:off + nb]; y = h[2][off:off + nb]
    eq = torch.equal(x, y)
    extra = ""
    if not eq:
        xf = x.view(torch.float32); yf = y.view(torch.float32)
        bad = (xf != yf).nonzero().flatten()
        extra = f"  {bad.numel()} of {xf.numel()} differ, first at {int(bad[0])}: {float(xf[bad[0]])} vs {float(yf[bad[0]])}, max abs diff {float((xf - yf).abs().max())}"
    print(f"{name:8s} {'same' if eq else 'DIFF'}{extra}")
print("timeouts", ops.chain_timeouts())
