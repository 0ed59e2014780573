"""Temporal-attention captioner: the host-side mirror of original_attention.py's
``Video_Caption_Generator`` (ctor :55-86, build_model :88-147, build_generator :176-251).

BASELINE.json names "attention_tf_s2vt score path"; attention_tf_s2vt.py itself holds no attention
op (SURVEY note N1) -- the arithmetic restated here is original_attention.py:95-134.  The forward
graph is composed from the C-ABI calls (s2vt_gemm for the projections, s2vt_attention_fwd for the
score/softmax/context, s2vt_lstm_cell_fwd for LSTM3); every activation is bit-identical to
oracle/s2vt_oracle.py::attention_forward.  Training of this variant is not wired yet (the kernels'
backward halves exist: s2vt_attention_bwd); see DESIGN.md "next".
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
        cur = torch.zeros(B, H, device=self.device)                                                           # (:105)
        w = p["embed_att_w"].view(-1).contiguous()
        vid = torch.arange(B, dtype=torch.int32, device=self.device); sid = torch.zeros(B, dtype=torch.int32, device=self.device)
        gsid = -torch.ones(B, dtype=torch.int32, device=self.device)
        logits = torch.empty(B, Tc, V, device=self.device); alphas = torch.empty(Tc, Tv, B, device=self.device)
        ids = torch.empty(B, Tc, dtype=torch.int32, device=self.device)
        if caption is not None:
            caption = torch.as_tensor(caption).to(self.device, torch.int32)
        for t in range(Tc):
            hWa = ops.gemm([op(h_prev)], p["embed_att_Wa"], None, M=B)
            _, alpha, ctx = ops.attention_fwd(hWa, P, Vt, w)                                                  # (:113-128)
            c, h, out, _ = ops.lstm_cell_fwd(op(ctx), op(cur), h_prev, c, p["lstm3_W"], p["lstm3_b"], B, keep=keep, seed=seed,
                                             video_id=vid, sample_id=sid, drop_code=768 + t)                  # (:131-132)
            y = ops.gemm([op(out), op(ctx), op(cur)], p["embed_nn_Wp"], p["embed_nn_bp"], M=B, act_tanh=True)  # (:134)
            h_prev = h
            tok, lg, _ = ops.vocab_pick(y, p["embed_word_W"], p["embed_word_b"], vid, gsid, t, 0, want_logits=True)  # (:143)
            logits[:, t] = lg
            alphas[t] = alpha
            ids[:, t] = tok
            nxt = tok if greedy else caption[:, t]
            cur = p["Wemb"][nxt.long()].contiguous()                                                          # (:141-142)
        return logits, alphas, (ids if greedy else None)
