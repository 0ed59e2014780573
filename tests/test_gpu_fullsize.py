"""BASELINE-size checks (B=64, K=5, Tc=20, d=1536, E=500, H=1000, |V|=12000) through size-independent
properties, plus an oracle spot check made affordable by batch-composition independence."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B, K, TC, TV, D, E, H, V = 64, 5, 20, 5, 1536, 500, 1000, 12000


@pytest.fixture(scope="module")
def full(gpu):
    import torch
    from s2vt_amd import model as M
    mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, TV, TC, seed=1234)
    g = torch.Generator().manual_seed(1)
    video = (torch.randn(B, TV, D, generator=g) * 0.5).abs().cuda()
    return mdl, video


def test_sampler_determinism_and_shard_independence(full, oracle):
    import torch
    mdl, video = full
    s1, g1 = mdl.sample(video, K, True, seed=99)
    s2, g2 = mdl.sample(video, K, True, seed=99)
    assert torch.equal(s1, s2) and torch.equal(g1, g2)                          # pure function of (weights, video, seed)
    s3, g3 = mdl.sample(video, K, True, seed=100)
    assert torch.equal(g1, g3) and not torch.equal(s1, s3)                      # greedy ignores the noise stream
    assert int(s1.min()) >= 0 and int(s1.max()) < V and s1.shape == (K * B, TC)
    # rank r of an 8-GPU run holds videos [8r, 8r+8): same tokens as the 1-GPU batch (global counters)
    lo, hi = 24, 32
    ss, gs = mdl.sample(video[lo:hi].contiguous(), K, True, seed=99, video_base=lo)
    full_rows = s1.view(K, B, TC)[:, lo:hi].reshape(-1, TC)
    assert torch.equal(ss, full_rows) and torch.equal(gs, g1[lo:hi])
    # oracle spot check on two videos of the full batch
    p = {n: mdl.store.p[n].cpu().numpy() for n in mdl.store.names}
    d = oracle.Dims(D, V, E, H, TV, TC, 0)
    rs, rg = oracle.sample_captions(p, d, video[10:12].cpu().numpy(), K, seed=99, video_base=10)
    assert np.array_equal(s1.view(K, B, TC)[:, 10:12].reshape(-1, TC).cpu().numpy(), rs)
    assert np.array_equal(g1[10:12].cpu().numpy(), rg)


def test_update_linearity_and_mask_normalisation(full):
    """The objective is linear in (r - b): gradients scale with it; dlogits rows sum to zero; scaling
    the mask-sum normaliser is exact.  lr = 0 keeps the variables fixed."""
    import torch
    mdl, video = full
    s, _ = mdl.sample(video, K, True, seed=5)
    is_eos = (s == 0)
    mask = ((torch.cumsum(is_eos.int(), 1) - is_eos.int()) == 0).float()
    g = torch.Generator().manual_seed(2)
    r = (torch.rand(K * B, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(K).cuda()
    step0 = mdl.global_step
    st1 = mdl.reinforce_update(video, s, mask, r, b, lr=0.0, clip_norm=5.0)
    g1 = mdl.store.grad[:mdl.store.numel].clone()
    dl = mdl._ctx[2]
    assert float(dl.sum(1).abs().max()) < 1e-4
    mdl.global_step = step0                                                     # same dropout masks
    st2 = mdl.reinforce_update(video, s, mask, 2 * r, 2 * b, lr=0.0, clip_norm=5.0)
    g2 = mdl.store.grad[:mdl.store.numel]
    assert torch.allclose(g2, 2 * g1, rtol=1e-3, atol=1e-7 * float(g1.abs().max()) + 1e-12)
    assert abs(float(st2.loss) - 2 * float(st1.loss)) < 1e-4 * abs(float(st1.loss)) + 1e-6
    assert abs(float(st2.grad_sumsq) - 4 * float(st1.grad_sumsq)) < 1e-2 * float(st1.grad_sumsq)
    assert float(st1.mask_sum) == float(mask.sum()) and float(mask[:, 0].min()) == 1.0
    mdl.global_step = step0


def test_fragment_order_decode_step_opt_in_bit_exact(gpu):
    """decode4.hip (opt-in, S2VT_DEC4=1: the sampler's LSTM2 step at 257-384 rows on fragment-order operands) draws
    the token ids of the default path, bit for bit, at the bench dimensions (B = 64, K = 5: R = 384) and at R = 272."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
from oracle import s2vt_oracle as orc
out = {}
for (B, K) in ((64, 5), (34, 7)):
    d = orc.Dims(1536, 12000, 500, 1000, 5, 6, 0)
    dims = ops.make_dims(1536, 12000, 500, 1000, 5, 6)
    p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 3).items()}
    video = torch.as_tensor(np.abs(np.random.default_rng(B).standard_normal((B, 5, 1536)) * 0.5).astype(np.float32)).cuda()
    s, g = ops.sample(dims, ops.make_params(p), video, K, seed=11)
    out[f"s{B}"] = s.cpu().numpy(); out[f"g{B}"] = g.cpu().numpy()
np.savez(sys.argv[1], **out)
print("child ok")
'''
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            f = os.path.join(td, f"ids{flag}.npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, S2VT_ROOT=root, S2VT_DEC4=flag), capture_output=True, text=True,
                               timeout=900)
            assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-3000:]
            res[flag] = dict(np.load(f))
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res["1"][k]), k


@pytest.mark.parametrize("mode", ["xe", "pg", "pg_live"])
def test_padding_steps_are_skipped_exactly_at_the_timed_shapes(full, mode):
    """active_steps at the bench shapes (persistent recurrences, split-K slabs, LDS-DMA weight gradients): the unroll
    stops behind the longest caption -- 13 of 20 steps here -- and loss and gradients are those of the full unroll (the
    skipped steps add zeros; gradients to the noise of the order-free reductions).  lr = 0 keeps the variables fixed."""
    import torch
    mdl, video = full
    rng = np.random.default_rng(3)
    live = mode == "pg_live"          # ... and, inside the unrolled steps, the vocabulary-sized kernels on the unmasked positions only
    rep = 1 if mode == "xe" else K
    N = rep * B
    longest = 13
    cap = rng.integers(2, V, (N, TC)).astype(np.int32)
    ln = rng.integers(1, longest, N); ln[5] = longest - 1
    for n in range(N):
        cap[n, ln[n]:] = 0
    mask = (np.arange(TC)[None, :] <= ln[:, None]).astype(np.float32)          # words + the first <eos>
    assert mdl.active_steps(mask) == longest
    r = (rng.random(N) * 2).astype(np.float32); b = np.tile((rng.random(B) * 2).astype(np.float32), rep)
    step0 = mdl.global_step
    outs = []
    for active in (None, "auto"):
        mdl.global_step = step0                                                 # same dropout masks
        if mode == "xe":
            st = mdl.xe_update(video, cap, mask, lr=0.0, q1=True, active_steps=active)
        else:
            st = mdl.reinforce_update(video, cap, mask, r, b, lr=0.0, active_steps=active, live_mask="auto" if (live and active) else None)
            assert (mdl._ctx[9] is not None) == bool(live and active)
        outs.append((float(st.loss), mdl.store.grad[:mdl.store.numel].clone(), float(st.grad_sumsq), mdl._ctx[8]))
    mdl.global_step = step0
    assert outs[0][3] == TC and outs[1][3] == longest
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * max(1.0, abs(outs[0][0]))
    scale = float(outs[0][1].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-5 * scale
    assert abs(outs[0][2] - outs[1][2]) <= 1e-4 * outs[0][2]
