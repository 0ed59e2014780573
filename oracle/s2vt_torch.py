"""Differentiable CPU restatement (torch, fp64 or fp32) -- TEST INFRASTRUCTURE, not product code.

Role 1: gradient / optimizer oracle.  The bit-exact forward contract lives in s2vt_oracle.c;
gradients, global-norm clip and TF-form Adam are checked against THIS file (autograd in
float64 over the same graph), within a stated tolerance.
Role 2: bench.py's ``cpu_baseline`` leg: the reference's step structure (K separate sampler
passes, one greedy pass, one forward/backward at K*B, clip, Adam) timed with torch-CPU fp32.

PARITY UNPINNED by the reference (TensorFlow 1.1 absent, no reference tests); see
s2vt_oracle.c.  Reference lines are cited per function (relative to /root/reference).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import torch


def to_torch(p, dtype=torch.float64, requires_grad=True):
    out = {}
    for k, v in p.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def lstm_cell(x, state_c, state_h, W, b, drop_mask=None, keep=1.0):
    """BasicLSTMCell + DropoutWrapper (tf_s2vt.py:74-77)."""
    H = state_h.shape[1]
    z = torch.cat([x, state_h], 1) @ W + b
    i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    c = state_c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    h = torch.tanh(c) * torch.sigmoid(o)
    out = h if drop_mask is None else (h / keep) * drop_mask
    return out, c, h


def unroll(p, video, prev_tokens_fn, Tc, drop=None, keep=1.0):
    """Encode (tf_s2vt.py:113-122) + decode (:126-153).  prev_tokens_fn(t, logits_prev) -> ids."""
    N, Tv, D = video.shape
    E = p["encode_image_W"].shape[1]
    H = p["lstm1_W"].shape[1] // 4
    dt = p["lstm1_W"].dtype
    emb = (video.reshape(N * Tv, D) @ p["encode_image_W"] + p["encode_image_b"]).reshape(N, Tv, E)
    z = lambda: torch.zeros(N, H, dtype=dt)
    c1, h1, c2, h2 = z(), z(), z(), z()
    pad = torch.zeros(N, E, dtype=dt)
    g = (lambda k, t: None) if drop is None else (lambda k, t: torch.as_tensor(drop[k][t]).to(dt))
    for t in range(Tv):
        o1, c1, h1 = lstm_cell(emb[:, t], c1, h1, p["lstm1_W"], p["lstm1_b"], g("enc1", t), keep)
        o2, c2, h2 = lstm_cell(torch.cat([o1, pad], 1), c2, h2, p["lstm2_W"], p["lstm2_b"], g("enc2", t), keep)
    logits_all = []
    logits = None
    for t in range(Tc):
        prev = prev_tokens_fn(t, logits)
        e = p["Wemb"][prev]
        o1, c1, h1 = lstm_cell(pad, c1, h1, p["lstm1_W"], p["lstm1_b"], g("dec1", t), keep)
        o2, c2, h2 = lstm_cell(torch.cat([o1, e], 1), c2, h2, p["lstm2_W"], p["lstm2_b"], g("dec2", t), keep)
        logits = o2 @ p["embed_word_W"] + p["embed_word_b"]
        logits_all.append(logits)
    return torch.stack(logits_all, 1)


def teacher_forced(p, video, caption, drop=None, keep=1.0):
    caption = torch.as_tensor(caption).long()
    N, Tc = caption.shape
    bos = torch.ones(N, dtype=torch.long)
    return unroll(p, video, lambda t, _: bos if t == 0 else caption[:, t - 1], Tc, drop, keep)


def xe_loss(p, logits, caption, mask, smoothing=0.05, loss_weight=1.0, decay=5e-5, q1=True, decay_all=False):
    """tf_s2vt.py:150-166 with TF-1.1 tf.losses.softmax_cross_entropy semantics.  decay_all: the always-true
    predicate of reinforce_multitask_e2e_attribute_s2vt.py:222 / e2e_tf_s2vt.py:199 -- the LSTM biases are decayed too."""
    caption = torch.as_tensor(caption).long()
    mask = torch.as_tensor(mask).to(logits.dtype)
    N, Tc, V = logits.shape
    lp = torch.log_softmax(logits, -1)
    q = torch.full_like(lp, smoothing / V)
    q.scatter_(2, caption.unsqueeze(-1), 1.0 - smoothing + smoothing / V)
    ce = -(q * lp).sum(-1)                                   # [N,Tc]
    if q1:
        tot = (ce.mean(0, keepdim=True) * mask).sum()
    else:
        tot = (ce * mask).sum()
    wd = sum(0.5 * (v ** 2).sum() for k, v in p.items() if decay_all or k not in ("lstm1_b", "lstm2_b"))
    return loss_weight * tot / mask.sum() + decay * wd


def pg_loss(logits, caption, mask, rewards, baseline):
    """reinforcement_multisampling_tf_s2vt.py:286-291,643-646."""
    caption = torch.as_tensor(caption).long()
    mask = torch.as_tensor(mask).to(logits.dtype)
    adv = (torch.as_tensor(rewards) - torch.as_tensor(baseline)).to(logits.dtype)
    lp = torch.log_softmax(logits, -1).gather(2, caption.unsqueeze(-1)).squeeze(-1)
    return -(lp * mask * adv[:, None]).sum() / mask.sum()


def attr_bce(p, video, labels, normalise=True):
    """reinforce_multitask_e2e_attribute_loss.py:375-380 (normalised) / :211-221 (plain sum)."""
    labels = torch.as_tensor(labels).to(video.dtype)
    z = video.mean(1) @ p["attr_W"] + p["attr_b"]
    bce = torch.clamp(z, min=0) - z * labels + torch.log1p(torch.exp(-z.abs()))
    s = bce.sum()
    return s / float(labels.shape[1] * labels.shape[0]) if normalise else s


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm."""
    n = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    s = clip / max(n, clip)
    return {k: g * s for k, g in grads.items()}, n


def adam_tf(p, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: eps outside the bias correction."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for k in p:
        m[k] = b1 * m[k] + (1 - b1) * grads[k]
        v[k] = b2 * v[k] + (1 - b2) * grads[k] * grads[k]
        p[k] = p[k] - lr_t * m[k] / (torch.sqrt(v[k]) + eps)
    return p, m, v


def exponential_decay(lr0, step, decay_steps, rate=0.5):
    """tf.train.exponential_decay(staircase=True) (App. B13)."""
    return lr0 * rate ** (step // decay_steps)


# ------------------------------------------------------------------------------------------
# attention model (original_attention.py:95-147), differentiable
# ------------------------------------------------------------------------------------------
def attention_teacher_forced(p, video, caption, drop=None, keep=1.0):
    caption = torch.as_tensor(caption).long()
    B, Tv, D = video.shape
    H = p["embed_att_Wa"].shape[0]
    Tc = caption.shape[1]
    dt = p["embed_att_Wa"].dtype
    Vt = (video.reshape(B * Tv, D) @ p["encode_image_W"] + p["encode_image_b"]).reshape(B, Tv, H).transpose(0, 1)
    P = Vt @ p["embed_att_Ua"] + p["embed_att_ba"]
    c = torch.zeros(B, H, dtype=dt); h_prev = torch.zeros(B, H, dtype=dt); emb = torch.zeros(B, H, dtype=dt)
    q_prev = torch.zeros(B, H, dtype=dt)        # attention query = the previous DropoutWrapper output (original_attention.py:135)
    out_logits, alphas = [], []
    for t in range(Tc):
        e = torch.tanh(q_prev @ p["embed_att_Wa"] + P) @ p["embed_att_w"]          # [Tv,B,1]
        ex = torch.exp(e.squeeze(-1))
        den = ex.sum(0)
        den = den + (den == 0).to(dt)
        alpha = ex / den
        ctx = (alpha.unsqueeze(-1) * Vt).sum(0)
        dm = None if drop is None else torch.as_tensor(drop[t]).to(dt)
        out, c, h = lstm_cell(torch.cat([ctx, emb], 1), c, h_prev, p["lstm3_W"], p["lstm3_b"], dm, keep)
        y = torch.tanh(torch.cat([out, ctx, emb], 1) @ p["embed_nn_Wp"] + p["embed_nn_bp"])
        h_prev = h
        q_prev = out
        emb = p["Wemb"][caption[:, t]]
        out_logits.append(y @ p["embed_word_W"] + p["embed_word_b"])
        alphas.append(alpha)
    return torch.stack(out_logits, 1), torch.stack(alphas, 0)


def attention_xe_loss(p, video, caption, mask, drop=None, keep=1.0, beta=10.0, m=0.5):
    """build_model's loss of the attention captioner (original_attention.py:136-150), differentiable:
        sum_{b,t} (ce[b,t] * mask[b,t] + beta * max(0, m - sum(alpha[t, 0:8, b])) * mask[b,t]) / sum(mask)
    (m = 0.5, beta = 10 at :299-300; the hinge is identically zero while Tv <= 8).  Returns (loss, logits, alphas)."""
    logits, alphas = attention_teacher_forced(p, video, caption, drop, keep)
    dt = logits.dtype
    mk = torch.as_tensor(mask).to(dt)
    lp = torch.log_softmax(logits, -1)
    ce = -lp.gather(2, torch.as_tensor(caption).long().unsqueeze(-1)).squeeze(-1)                 # [B,Tc]
    s = alphas[:, :8, :].sum(1)                                                                    # [Tc,B]
    reg = beta * torch.clamp(m - s, min=0.0).transpose(0, 1) * mk
    return ((ce * mk).sum() + reg.sum()) / mk.sum(), logits, alphas


# ------------------------------------------------------------------------------------------
# cpu_baseline: the reference's REINFORCE step structure with torch-CPU fp32
# (reinforcement_multisampling_tf_s2vt.py:743-753 sampling, :823-826 update)
# ------------------------------------------------------------------------------------------
def reference_structured_step(p, m, v, step, video, K, Tc, rewards, baseline, lr0=1e-6, clip=5.0, keep=0.9,
                              gen: torch.Generator | None = None):
    """One REINFORCE step as the reference schedules it: K multinomial sampler passes and one
    greedy pass (each re-encoding the video), host-side mask building, then a dropout-wrapped
    forward/backward at N=K*B, clip_by_global_norm(5), TF-form Adam.  Returns sampled ids."""
    B = video.shape[0]
    dt = video.dtype
    samples = []
    with torch.no_grad():
        for k in range(K + 1):
            ids = []

            def pick(t, logits, greedy=(k == K)):
                if t == 0:
                    return torch.ones(B, dtype=torch.long)
                if greedy:
                    tok = logits.argmax(1)
                else:
                    tok = torch.multinomial(torch.softmax(logits, 1), 1, generator=gen).squeeze(1)
                ids.append(tok)
                return tok

            logits = unroll(p, video, pick, Tc)
            last = logits[:, -1]
            ids.append(last.argmax(1) if k == K else torch.multinomial(torch.softmax(last, 1), 1, generator=gen).squeeze(1))
            samples.append(torch.stack(ids, 1))
    sampled = torch.cat(samples[:K], 0)                        # sample-major (:764-782)
    # mask = 1 up to and including the first <eos>=0 (cider_evaluation.py:145-172)
    is_eos = (sampled == 0)
    seen = torch.cumsum(is_eos.int(), 1) - is_eos.int()
    mask = (seen == 0).to(dt)
    vid_t = video.repeat(K, 1, 1)
    N = K * B
    H = p["lstm1_W"].shape[1] // 4
    Tv = video.shape[1]
    mk = lambda T: (torch.rand(T, N, H, generator=gen) < keep).to(dt)
    drop = {"enc1": mk(Tv), "enc2": mk(Tv), "dec1": mk(Tc), "dec2": mk(Tc)}
    for t_ in p.values():
        t_.grad = None
    logits = teacher_forced(p, vid_t, sampled, drop, keep)
    loss = pg_loss(logits, sampled, mask, rewards, baseline)
    loss.backward()
    with torch.no_grad():
        grads = {k: t_.grad for k, t_ in p.items()}
        grads, _ = clip_by_global_norm(grads, clip)
        lr = exponential_decay(lr0, step, 1000)
        lr_t = lr * math.sqrt(1.0 - 0.999 ** (step + 1)) / (1.0 - 0.9 ** (step + 1))
        for k in p:
            m[k].mul_(0.9).add_(grads[k], alpha=0.1)
            v[k].mul_(0.999).addcmul_(grads[k], grads[k], value=0.001)
            p[k].sub_(lr_t * m[k] / (v[k].sqrt() + 1e-8))
    return sampled, float(loss)
