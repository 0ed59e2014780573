"""The oracle's building blocks against THIRD-PARTY implementations that ship with the image (torch.nn / torch.optim), so
that the cell, the losses, the clip and the optimizer are not one person's recall of TensorFlow 1.1 (SURVEY App. B2, B5,
B11, B12, B14).  CPU only.  This does not pin parity to the reference (nothing here can: TF 1.1 is absent) -- it pins the
restatement to independent code wherever the two SHOULD agree, and shows the documented differences explicitly."""
import numpy as np
import torch

from oracle import s2vt_torch as T


def _rng(seed=0):
    return torch.Generator().manual_seed(seed)


def test_basic_lstm_cell_vs_torch_lstmcell():
    """BasicLSTMCell (gate order i, j, f, o; forget_bias 1.0 added at run time; W rows [x ; h]) == torch.nn.LSTMCell (gate
    order i, f, g, o; W_ih / W_hh as [4H, in]) after permuting the gate blocks and folding +1 into the forget bias."""
    g = _rng(1)
    B, X, H = 7, 5, 6
    W = torch.randn(X + H, 4 * H, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.3
    x = torch.randn(B, X, generator=g, dtype=torch.float64)
    c = torch.randn(B, H, generator=g, dtype=torch.float64)
    h = torch.randn(B, H, generator=g, dtype=torch.float64)
    _, c1, h1 = T.lstm_cell(x, c, h, W, b)
    cell = torch.nn.LSTMCell(X, H, dtype=torch.float64)
    blk = lambda M, k: M[..., k * H:(k + 1) * H]
    perm = (0, 2, 1, 3)                                  # torch order (i, f, g, o) <- ours (i, j, f, o): blocks 0, 2, 1, 3
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([blk(W[:X], k) for k in perm], 1).t())
        cell.weight_hh.copy_(torch.cat([blk(W[X:], k) for k in perm], 1).t())
        bias = torch.cat([blk(b, k) for k in perm]).clone()
        bias[H:2 * H] += 1.0                             # forget_bias = 1.0 lives in the graph, not in the variable
        cell.bias_ih.copy_(bias)
        cell.bias_hh.zero_()
    h2, c2 = cell(x, (h, c))
    assert torch.allclose(h1, h2, atol=1e-12) and torch.allclose(c1, c2, atol=1e-12)
    # and the C oracle's fp32 pointwise sequence agrees with both to fp32 accuracy
    from oracle import s2vt_oracle as orc
    z = (torch.cat([x, h], 1) @ W + b).float().numpy()
    c3, h3, _, _ = orc.lstm_pointwise(np.ascontiguousarray(z), c.float().numpy())
    assert np.abs(c3 - c2.detach().numpy()).max() < 2e-6 and np.abs(h3 - h2.detach().numpy()).max() < 2e-6


def test_smoothed_cross_entropy_vs_torch_functional():
    """q = onehot * 0.95 + 0.05 / V; ce = -sum q log_softmax (tf_s2vt.py:150-155) == F.cross_entropy(label_smoothing=0.05) per row;
    the oracle's fp32 row_losses agrees; Q1 (TF-1.1 losses return the batch mean) is the one place ours differs, shown explicitly."""
    import torch.nn.functional as F
    from oracle import s2vt_oracle as orc
    g = _rng(2)
    N, Tc, V = 5, 4, 37
    logits = torch.randn(N, Tc, V, generator=g, dtype=torch.float64) * 2
    cap = torch.randint(0, V, (N, Tc), generator=g)
    mask = (torch.rand(N, Tc, generator=g) < 0.7).double(); mask[:, 0] = 1
    ce = F.cross_entropy(logits.reshape(-1, V), cap.reshape(-1), reduction="none", label_smoothing=0.05).reshape(N, Tc)
    p = {"w": torch.zeros(1, dtype=torch.float64)}                                       # (no weight decay term)
    ours = T.xe_loss(p, logits, cap.numpy(), mask.numpy(), smoothing=0.05, decay=0.0, q1=False)
    assert abs(float(ours) - float((ce * mask).sum() / mask.sum())) < 1e-12
    q1 = T.xe_loss(p, logits, cap.numpy(), mask.numpy(), smoothing=0.05, decay=0.0, q1=True)
    assert abs(float(q1) - float((ce.mean(0, keepdim=True) * mask).sum() / mask.sum())) < 1e-12       # batch MEAN times the mask (Q1)
    for t in range(Tc):
        nll, lp, _ = orc.row_losses(np.ascontiguousarray(logits[:, t].float().numpy()), cap[:, t].numpy().astype(np.int32), 0.05)
        assert np.abs(nll - ce[:, t].numpy()).max() < 5e-6
        plain = F.cross_entropy(logits[:, t], cap[:, t], reduction="none")
        assert np.abs(-lp - plain.numpy()).max() < 5e-6                                  # lp = log-probability of the target


def test_policy_gradient_loss_vs_torch_nll():
    import torch.nn.functional as F
    g = _rng(3)
    N, Tc, V = 6, 5, 23
    logits = torch.randn(N, Tc, V, generator=g, dtype=torch.float64)
    cap = torch.randint(0, V, (N, Tc), generator=g)
    mask = (torch.rand(N, Tc, generator=g) < 0.6).double(); mask[:, 0] = 1
    r = torch.rand(N, generator=g, dtype=torch.float64); b = torch.rand(N, generator=g, dtype=torch.float64)
    nll = F.nll_loss(F.log_softmax(logits, -1).reshape(-1, V), cap.reshape(-1), reduction="none").reshape(N, Tc)
    ref = (nll * mask * (r - b)[:, None]).sum() / mask.sum()
    assert abs(float(T.pg_loss(logits, cap.numpy(), mask.numpy(), r, b)) - float(ref)) < 1e-12


def test_sigmoid_bce_vs_torch_functional():
    """max(z, 0) - z y + log1p(exp(-|z|)) (tf.nn.sigmoid_cross_entropy_with_logits, App. B11) == F.binary_cross_entropy_with_logits."""
    import torch.nn.functional as F
    from oracle import s2vt_oracle as orc
    g = _rng(4)
    B, Tv, D, A = 4, 3, 6, 9
    p = {"attr_W": torch.randn(D, A, generator=g, dtype=torch.float64), "attr_b": torch.randn(A, generator=g, dtype=torch.float64)}
    video = torch.rand(B, Tv, D, generator=g, dtype=torch.float64) * 3
    y = (torch.rand(B, A, generator=g) < 0.3).double()
    z = video.mean(1) @ p["attr_W"] + p["attr_b"]
    ref = F.binary_cross_entropy_with_logits(z, y, reduction="sum")
    assert abs(float(T.attr_bce(p, video, y.numpy(), normalise=False)) - float(ref)) < 1e-10
    assert abs(float(T.attr_bce(p, video, y.numpy(), normalise=True)) - float(ref) / (A * B)) < 1e-12
    _, bce = orc.attr_head({k: v.float().numpy() for k, v in p.items()}, video.float().numpy(), y.float().numpy())
    assert abs(float(bce.astype(np.float64).sum()) - float(ref)) < 1e-4 * float(ref)


def test_clip_by_global_norm_vs_torch_utils():
    """tf.clip_by_global_norm: g * clip / max(norm, clip) == torch.nn.utils.clip_grad_norm_ (which divides by norm + 1e-6: shown)."""
    g = _rng(5)
    shapes = [(3, 4), (7,), (2, 2, 2)]
    for scale, clip in ((10.0, 5.0), (0.01, 5.0)):
        grads = {f"v{i}": torch.randn(*s, generator=g, dtype=torch.float64) * scale for i, s in enumerate(shapes)}
        ours, n = T.clip_by_global_norm(grads, clip)
        ps = [torch.nn.Parameter(torch.zeros(*s, dtype=torch.float64)) for s in shapes]
        for p_, gr in zip(ps, grads.values()):
            p_.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_(ps, clip)
        assert abs(float(tn) - n) < 1e-12 * max(1.0, n)
        for p_, k in zip(ps, grads):
            # torch scales by clip / (norm + 1e-6) (clamped to 1); TF by clip / max(norm, clip): equal up to that epsilon
            assert torch.allclose(p_.grad, ours[k], rtol=2e-6, atol=0)


def test_tf_adam_vs_torch_optim_adam_and_the_epsilon_placement():
    """TF-form Adam (Q6): theta -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)  -- eps OUTSIDE the bias correction.
    torch.optim.Adam: theta -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The two coincide exactly when torch is
    given eps_torch = eps / sqrt(1 - b2^t) at step t; with the same eps they differ where sqrt(v) ~ eps -- both facts checked."""
    g = _rng(6)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    theta0 = torch.randn(50, generator=g, dtype=torch.float64)
    grads = [torch.randn(50, generator=g, dtype=torch.float64) * (10.0 ** -i) for i in (0, 3, 9)]      # down to |g| ~ eps
    p = {"w": theta0.clone()}
    m = {"w": torch.zeros_like(theta0)}
    v = {"w": torch.zeros_like(theta0)}
    w = torch.nn.Parameter(theta0.clone())
    opt = torch.optim.Adam([w], lr=lr, betas=(b1, b2), eps=eps)
    w_same_eps = torch.nn.Parameter(theta0.clone())
    opt_same = torch.optim.Adam([w_same_eps], lr=lr, betas=(b1, b2), eps=eps)
    for t, gr in enumerate(grads, 1):
        p, m, v = T.adam_tf(p, {"w": gr}, m, v, t, lr, b1, b2, eps)
        for group in opt.param_groups:
            group["eps"] = eps / np.sqrt(1.0 - b2 ** t)
        w.grad = gr.clone(); opt.step()
        w_same_eps.grad = gr.clone(); opt_same.step()
        assert torch.allclose(p["w"], w.detach(), rtol=0, atol=1e-15)
    assert float((p["w"] - w_same_eps.detach()).abs().max()) > 1e-12          # same eps, different placement: NOT the same optimizer


def test_exponential_decay_staircase_vs_torch_steplr():
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([w], lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.5)
    for step in range(23):
        assert abs(opt.param_groups[0]["lr"] - T.exponential_decay(1e-3, step, 5)) < 1e-18
        opt.step(); sched.step()


def test_attention_softmax_context_vs_torch():
    """alpha = exp(e) / sum_t exp(e) without a max shift (original_attention.py:116-121) == softmax over frames wherever exp does not
    overflow; ctx = sum_t alpha V (einsum).  The C oracle's fp32 step agrees."""
    from oracle import s2vt_oracle as orc
    g = _rng(7)
    Tv, B, H = 6, 4, 12
    hWa = torch.randn(B, H, generator=g); P = torch.randn(Tv, B, H, generator=g); V = torch.randn(Tv, B, H, generator=g)
    w = torch.rand(H, generator=g) * 0.2 - 0.1
    e = torch.einsum("tbh,h->tb", torch.tanh(hWa.double() + P.double()), w.double())
    alpha = torch.softmax(e, 0)
    ctx = torch.einsum("tb,tbh->bh", alpha, V.double())
    a2, c2 = orc.attention_step(hWa.numpy(), P.numpy(), V.numpy(), w.numpy())
    assert np.abs(a2 - alpha.numpy()).max() < 2e-6 and np.abs(c2 - ctx.numpy()).max() < 5e-6
