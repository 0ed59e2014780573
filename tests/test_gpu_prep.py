"""s2vt_xe_prep / s2vt_mixed_prep / s2vt_mixed_loss: the one-launch forms of the coefficient preparation of the XE update
(tf_s2vt.py:150-166) and of the mixed multitask objective (reinforce_multitask_e2e_attribute_s2vt.py:850) against the tensor-library
expressions they replace (model.py keeps those for data-parallel runs): coefficients, targets and sums bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Tc,q1", [(64, 20, True), (64, 20, False), (7, 5, True), (130, 35, False)])
def test_xe_prep_equals_the_tensor_expressions(gpu, N, Tc, q1):
    import torch
    rng = np.random.default_rng(N + Tc)
    ln = rng.integers(1, Tc + 1, N)
    mask = torch.as_tensor((np.arange(Tc)[None, :] < ln[:, None]).astype(np.float32)).cuda()
    cap = torch.as_tensor(rng.integers(0, 1000, (N, Tc)).astype(np.int32)).cuda()
    lw, n_glob = 1.0, float(N)
    coef, tgt, msum = gpu.xe_prep(mask, cap, lw, n_glob, q1)
    colsum = mask.sum(0)
    ref = ((colsum[:, None] / n_glob).expand(-1, N) * lw) if q1 else (mask.t() * lw)
    assert torch.equal(coef, ref.contiguous().view(-1))
    assert torch.equal(tgt, cap.t().contiguous().view(-1))
    assert float(msum) == float(mask.sum())


@pytest.mark.parametrize("lam", [0.5, 0.1, 0.7, 0.9])          # non-dyadic: 1 - lambda must be rounded ONCE, from the double
@pytest.mark.parametrize("B,rep,Tc,q1", [(32, 1, 20, True), (32, 1, 20, False), (5, 3, 7, True), (6, 2, 9, True)])
def test_mixed_prep_and_loss_equal_the_tensor_expressions(gpu, B, rep, Tc, q1, lam):
    import torch
    rng = np.random.default_rng(B * 10 + rep)
    Ns, N = B * rep, B * (rep + 1)
    mk = lambda n: torch.as_tensor((np.arange(Tc)[None, :] < rng.integers(1, Tc + 1, n)[:, None]).astype(np.float32)).cuda()
    mask, gmask = mk(Ns), mk(B)
    r = torch.as_tensor(rng.random(Ns).astype(np.float32) * 2).cuda(); b = torch.as_tensor(rng.random(Ns).astype(np.float32)).cuda()
    cap = torch.as_tensor(rng.integers(0, 1000, (Ns, Tc)).astype(np.int32)).cuda()
    gcap = torch.as_tensor(rng.integers(0, 1000, (B, Tc)).astype(np.int32)).cuda()
    lw, sm = 1.0, 0.05
    coef, smooth, cap_all, tgt, sums = gpu.mixed_prep(mask, gmask, r, b, cap, gcap, lam, lw, q1, sm, float(B))
    s = torch.stack([mask.sum(), gmask.sum()])
    one_minus = torch.full((), 1.0 - lam, dtype=torch.float32, device="cuda")       # model.mixed_update's data-parallel branch, verbatim
    lam_t = torch.full((), lam, dtype=torch.float32, device="cuda")
    coef_pg = mask * ((r - b) * one_minus)[:, None] / s[0]
    coef_xe = ((gmask.sum(0)[None, :] / float(B)).expand(B, -1) * lw) if q1 else (gmask * lw)
    coef_xe = coef_xe * (lam_t / s[1])
    ref = torch.cat([coef_pg, coef_xe], 0).t().contiguous().view(-1)
    assert torch.equal(sums, s)
    assert torch.equal(coef, ref)
    sref = torch.zeros(N, device="cuda"); sref[Ns:] = sm
    assert torch.equal(smooth, sref.repeat(Tc))
    assert torch.equal(cap_all, torch.cat([cap, gcap], 0))
    assert torch.equal(tgt, cap_all.t().contiguous().view(-1))
    # loss terms: dense rows and a live list
    nll = torch.as_tensor(rng.random(N * Tc).astype(np.float32)).cuda()
    out = gpu.mixed_loss(coef, nll, None, N, Ns)
    per_row = (coef * nll).view(-1, N)
    want = torch.stack([per_row[:, :Ns].sum(), per_row[:, Ns:].sum()])
    assert torch.allclose(out[:2], want, rtol=2e-6, atol=1e-7) and abs(float(out[2]) - float(want.sum())) <= 2e-6 * max(1.0, abs(float(want.sum())))
    live = torch.as_tensor(np.flatnonzero(rng.random(N * Tc) < 0.4).astype(np.int32)).cuda()
    cl, nl = coef[live.long()].contiguous(), nll[live.long()].contiguous()
    out = gpu.mixed_loss(cl, nl, live, N, Ns)
    is_pg = (live.long() % N) < Ns
    want = torch.stack([(cl * nl)[is_pg].sum(), (cl * nl)[~is_pg].sum()])
    assert torch.allclose(out[:2], want, rtol=2e-6, atol=1e-7)
