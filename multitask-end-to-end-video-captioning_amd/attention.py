"""Temporal-attention captioner: the host-side mirror of original_attention.py's
``Video_Caption_Generator`` (ctor :55-86, build_model :88-147, build_generator :176-251).

BASELINE.json names "attention_tf_s2vt score path"; attention_tf_s2vt.py itself holds no attention
op (SURVEY note N1) -- the arithmetic restated here is original_attention.py:95-134.  The forward
graph is composed from the C-ABI calls (s2vt_gemm for the projections, s2vt_attention_fwd for the
score/softmax/context, s2vt_lstm_cell_fwd for LSTM3); every activation is bit-identical to
oracle/s2vt_oracle.py::attention_forward.  ``xe_update`` is the training step of the variant
(build_model's loss :136-147 + the train_op of its train(): Adam on the gradients): the backward graph is
composed on the host from the library's order-free pieces (s2vt_attention_bwd, s2vt_lstm_cell_bwd, s2vt_gemm /
s2vt_gemm_tn, s2vt_softmax_nll_fwd_bwd, ...), checked against float64 autograd of the torch restatement.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

NAMES = ("Wemb", "encode_image_W", "encode_image_b", "embed_att_w", "embed_att_Wa", "embed_att_Ua", "embed_att_ba",
         "embed_word_W", "embed_word_b", "embed_nn_Wp", "embed_nn_bp", "lstm3_W", "lstm3_b")
TF_NAMES = {n: n for n in NAMES}
TF_NAMES.update({"lstm3_W": "s2vt/LSTM3/basic_lstm_cell/weights", "lstm3_b": "s2vt/LSTM3/basic_lstm_cell/biases"})


class Attention_Caption_Generator:
    def __init__(self, dim_image, n_words, dim_hidden, batch_size, n_video_lstm_steps, n_caption_lstm_steps, drop_out_rate,
                 bias_init_vector=None, device="cuda", seed=1234):
        self.dim_image, self.n_words, self.dim_hidden, self.batch_size = dim_image, n_words, dim_hidden, batch_size
        self.n_video_lstm_steps, self.n_caption_lstm_steps, self.drop_out_rate = n_video_lstm_steps, n_caption_lstm_steps, drop_out_rate
        self.device = torch.device(device)
        H, V, D = dim_hidden, n_words, dim_image
        g = torch.Generator().manual_seed(seed)
        u = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.1).to(self.device)
        a = float(np.sqrt(6.0 / (3 * H + 4 * H)))
        self.p = {
            "Wemb": u(V, H), "encode_image_W": u(D, H), "encode_image_b": torch.zeros(H, device=self.device),
            "embed_att_w": u(H, 1), "embed_att_Wa": u(H, H), "embed_att_Ua": u(H, H),
            "embed_att_ba": torch.zeros(H, device=self.device), "embed_word_W": u(H, V),
            "embed_word_b": torch.zeros(V, device=self.device), "embed_nn_Wp": u(3 * H, H),
            "embed_nn_bp": torch.zeros(H, device=self.device),
            "lstm3_W": ((torch.rand(3 * H, 4 * H, generator=g) * 2 - 1) * a).to(self.device),
            "lstm3_b": torch.zeros(4 * H, device=self.device),
        }
        if bias_init_vector is not None:
            self.p["embed_word_b"].copy_(torch.as_tensor(np.asarray(bias_init_vector, np.float32)))

    def load(self, arrays):
        for k, v in arrays.items():
            if k in self.p:
                self.p[k].copy_(torch.as_tensor(np.asarray(v, np.float32)).to(self.device))

    def forward(self, video, caption=None, greedy=False, keep=1.0, seed=0):
        """Teacher-forced logits [B,Tc,V] + alphas [Tc,Tv,B] (build_model, :95-147), or the greedy ids
        (build_generator) when greedy=True.  LSTM3 output dropout as DropoutWrapper (:78) when keep<1."""
        p = self.p
        video = torch.as_tensor(video).to(self.device, torch.float32).contiguous()
        B, Tv, D = video.shape
        H, V, Tc = self.dim_hidden, self.n_words, self.n_caption_lstm_steps
        op = ops.operand
        emb = ops.gemm([op(video.view(B * Tv, D))], p["encode_image_W"], p["encode_image_b"], M=B * Tv)      # (b,t) rows
        Vt = emb.view(B, Tv, H).transpose(0, 1).contiguous()                                                  # [Tv,B,H] (:98)
        P = ops.gemm([op(Vt.view(Tv * B, H))], p["embed_att_Ua"], p["embed_att_ba"], M=Tv * B).view(Tv, B, H)  # (:107)
        c = torch.zeros(B, H, device=self.device); h_prev = torch.zeros(B, H, device=self.device)
        q_prev = torch.zeros(B, H, device=self.device)          # attention query: the previous DropoutWrapper output (:135)
        cur = torch.zeros(B, H, device=self.device)                                                           # (:105)
        w = p["embed_att_w"].view(-1).contiguous()
        vid = torch.arange(B, dtype=torch.int32, device=self.device); sid = torch.zeros(B, dtype=torch.int32, device=self.device)
        gsid = -torch.ones(B, dtype=torch.int32, device=self.device)
        logits = torch.empty(B, Tc, V, device=self.device); alphas = torch.empty(Tc, Tv, B, device=self.device)
        ids = torch.empty(B, Tc, dtype=torch.int32, device=self.device)
        if caption is not None:
            caption = torch.as_tensor(caption).to(self.device, torch.int32)
        for t in range(Tc):
            hWa = ops.gemm([op(q_prev)], p["embed_att_Wa"], None, M=B)
            _, alpha, ctx = ops.attention_fwd(hWa, P, Vt, w)                                                  # (:113-128)
            c, h, out, _ = ops.lstm_cell_fwd(op(ctx), op(cur), h_prev, c, p["lstm3_W"], p["lstm3_b"], B, keep=keep, seed=seed,
                                             video_id=vid, sample_id=sid, drop_code=768 + t)                  # (:131-132)
            y = ops.gemm([op(out), op(ctx), op(cur)], p["embed_nn_Wp"], p["embed_nn_bp"], M=B, act_tanh=True)  # (:134)
            h_prev = h
            q_prev = out
            tok, lg, _ = ops.vocab_pick(y, p["embed_word_W"], p["embed_word_b"], vid, gsid, t, 0, want_logits=True)  # (:143)
            logits[:, t] = lg
            alphas[t] = alpha
            ids[:, t] = tok
            nxt = tok if greedy else caption[:, t]
            cur = p["Wemb"][nxt.long()].contiguous()                                                          # (:141-142)
        return logits, alphas, (ids if greedy else None)

    # ------------------------------------------------------------------------------------------ training
    def xe_update(self, video, caption, caption_mask, lr, clip_norm=0.0, keep=None, seed=0, beta1=0.9, beta2=0.999, eps=1e-8):
        """One training step of the attention captioner: loss = sum_{b,t} ce[b,t]*mask[b,t] / sum(mask)
        (original_attention.py:136-147; the alpha regulariser beta*max(0, m - sum(alpha[:, :8])) with m = 0.5 is
        identically zero while n_video_lstm_steps <= 8 -- asserted), gradients by BPTT through the unroll,
        optional global-norm clip, TF-form Adam.  Returns (loss, grads dict)."""
        p = self.p
        assert self.n_video_lstm_steps <= 8, "the alpha regulariser (original_attention.py:123,146) is only zero for <= 8 frames"
        keep = self.drop_out_rate if keep is None else keep
        dev = self.device
        video = torch.as_tensor(video).to(dev, torch.float32).contiguous()
        cap = torch.as_tensor(caption).to(dev, torch.int32).contiguous()
        mask = torch.as_tensor(caption_mask).to(dev, torch.float32).contiguous()
        B, Tv, D = video.shape
        H, V, Tc = self.dim_hidden, self.n_words, self.n_caption_lstm_steps
        op = ops.operand
        vid = torch.arange(B, dtype=torch.int32, device=dev); sid = torch.zeros(B, dtype=torch.int32, device=dev)
        # ---- forward, keeping what the backward needs
        emb = ops.gemm([op(video.view(B * Tv, D))], p["encode_image_W"], p["encode_image_b"], M=B * Tv)
        Vt = emb.view(B, Tv, H).transpose(0, 1).contiguous()
        P = ops.gemm([op(Vt.view(Tv * B, H))], p["embed_att_Ua"], p["embed_att_ba"], M=Tv * B).view(Tv, B, H)
        w = p["embed_att_w"].view(-1).contiguous()
        z0 = torch.zeros(B, H, device=dev)
        c, h_prev, q_prev, cur = z0, z0, z0, z0
        sv = []
        logits = torch.empty(Tc * B, V, device=dev)
        for t in range(Tc):
            hWa = ops.gemm([op(q_prev)], p["embed_att_Wa"], None, M=B)
            _, alpha, ctx = ops.attention_fwd(hWa, P, Vt, w)
            c_new, h, out, gates = ops.lstm_cell_fwd(op(ctx), op(cur), h_prev, c, p["lstm3_W"], p["lstm3_b"], B, keep=keep, seed=seed,
                                                     video_id=vid, sample_id=sid, drop_code=768 + t, want_gates=True)
            y = ops.gemm([op(out), op(ctx), op(cur)], p["embed_nn_Wp"], p["embed_nn_bp"], M=B, act_tanh=True)
            ops.gemm([op(y)], p["embed_word_W"], p["embed_word_b"], M=B, out=logits[t * B:(t + 1) * B])
            sv.append((hWa, alpha, ctx, c, c_new, gates, out, y, cur, q_prev, h_prev))
            c, h_prev, q_prev = c_new, h, out
            cur = p["Wemb"][cap[:, t].long()].contiguous()
        msum = mask.sum()
        coef = mask.t().contiguous().view(-1)                         # time-major [Tc*B]
        nll, _ = ops.softmax_nll_fwd_bwd(logits, cap.t().contiguous().view(-1), coef, 0.0)      # logits <- d/dlogits (un-normalised)
        loss = torch.dot(coef, nll) / msum
        # ---- backward
        g = {k: torch.zeros_like(v) for k, v in p.items()}
        WoutT, WpT, W3T = ops.transpose(p["embed_word_W"]), ops.transpose(p["embed_nn_Wp"]), ops.transpose(p["lstm3_W"])
        WaT, UaT = ops.transpose(p["embed_att_Wa"]), ops.transpose(p["embed_att_Ua"])
        dP_tot = torch.zeros_like(P); dVt_tot = torch.zeros_like(Vt)
        dw = torch.zeros(H, device=dev)
        dq_next = torch.zeros(B, H, device=dev)       # gradient w.r.t. this step's dropped output from the NEXT step's attention query
        dh_rec = torch.zeros(B, H, device=dev)        # ... w.r.t. this step's clean h from the next step's LSTM
        dc = None
        for t in range(Tc - 1, -1, -1):
            hWa, alpha, ctx, c_prev, c_new, gates, out, y, cur, q_prev, h_prev = sv[t]
            dl = logits[t * B:(t + 1) * B]
            ops.gemm_tn(y, dl, g["embed_word_W"]); ops.colsum(dl, g["embed_word_b"])
            dy = ops.gemm([op(dl)], WoutT, None, M=B)
            dpre = ops.tanh_bwd(y, dy)
            ops.gemm_tn(out, dpre, g["embed_nn_Wp"][:H]); ops.gemm_tn(ctx, dpre, g["embed_nn_Wp"][H:2 * H])
            ops.gemm_tn(cur, dpre, g["embed_nn_Wp"][2 * H:]); ops.colsum(dpre, g["embed_nn_bp"])
            dcat = ops.gemm([op(dpre)], WpT, None, M=B)                                   # d[out ; ctx ; cur]
            dout = dcat[:, :H] + dq_next
            dh = ops.dropout_bwd(dout.contiguous(), keep, seed, 768 + t, vid, sid) + dh_rec
            dz, dc = ops.lstm_cell_bwd(gates, c_new, c_prev, dh.contiguous(), dc)
            ops.gemm_tn(ctx, dz, g["lstm3_W"][:H]); ops.gemm_tn(cur, dz, g["lstm3_W"][H:2 * H])
            ops.gemm_tn(h_prev, dz, g["lstm3_W"][2 * H:]); ops.colsum(dz, g["lstm3_b"])
            dx = ops.gemm([op(dz)], W3T, None, M=B)                                       # d[ctx ; cur ; h_prev]
            dh_rec = dx[:, 2 * H:].contiguous()
            dctx = (dcat[:, H:2 * H] + dx[:, :H]).contiguous()
            dcur = (dcat[:, 2 * H:] + dx[:, H:2 * H]).contiguous()
            if t > 0:
                ops.lib().s2vt_embed_scatter_add(ops._ptr(dcur), dcur.stride(0), ops._ptr(cap[:, t - 1].contiguous()), B, H,
                                                 ops._ptr(g["Wemb"]), ops._stream())
            dhWa, dP, dVt = ops.attention_bwd(hWa, P, Vt, w, alpha, dctx, dw)
            dP_tot += dP; dVt_tot += dVt
            ops.gemm_tn(q_prev, dhWa, g["embed_att_Wa"])
            dq_next = ops.gemm([op(dhWa)], WaT, None, M=B)
        g["embed_att_w"] += dw.view(H, 1)
        dPf = dP_tot.view(Tv * B, H)
        ops.gemm_tn(Vt.view(Tv * B, H), dPf, g["embed_att_Ua"]); ops.colsum(dPf, g["embed_att_ba"])
        dVt_tot += ops.gemm([op(dPf)], UaT, None, M=Tv * B).view(Tv, B, H)
        demb = dVt_tot.transpose(0, 1).contiguous().view(B * Tv, H)                       # back to (b, t) rows
        ops.gemm_tn(video.view(B * Tv, D), demb, g["encode_image_W"]); ops.colsum(demb, g["encode_image_b"])
        inv = (1.0 / msum)
        for k in g:
            g[k] *= inv
        # ---- clip + TF-form Adam (same kernels as the S2VT trainer, one call per variable)
        if not hasattr(self, "_m"):
            self._m = {k: torch.zeros_like(v) for k, v in p.items()}
            self._v = {k: torch.zeros_like(v) for k, v in p.items()}
            self._step = 0
        self._step += 1
        sumsq = torch.zeros(1, device=dev)
        for k in g:
            sumsq += (g[k].double() ** 2).sum().float()
        for k in p:
            ops.adam_tf(p[k].view(-1), g[k].view(-1), self._m[k].view(-1), self._v[k].view(-1), sumsq, clip_norm, lr, self._step,
                        beta1, beta2, eps)
        return loss, g
