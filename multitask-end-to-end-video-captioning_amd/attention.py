"""Temporal-attention captioner: the host-side mirror of original_attention.py's ``Video_Caption_Generator``
(ctor :55-86, build_model :88-150, build_generator :155-199, build_sampler :201-251, the train_op of train() :430-441).

BASELINE.json names "attention_tf_s2vt score path"; attention_tf_s2vt.py itself holds no attention op (SURVEY note N1) --
the arithmetic restated here is original_attention.py:95-147.  The class keeps the reference constructor and the
``build_*`` surface: the methods return the same tuples of placeholders / fetches, ``model.Session(m).run(fetches, feed)``
evaluates them, and ``exponential_decay`` / ``minimize`` stand for the nodes the reference's train() adds (:430-441), so
that loop translates statement by statement (tests/test_gpu_attention_model.py replays it).  All arithmetic is in
libs2vt_hip.so (csrc/attn_model.hip: the whole unroll, its backward and the greedy decode loop are library calls; csrc/attn.hip:
score -> softmax -> context in one launch per step); torch supplies device memory and streams.  No CPU fallback.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import dist as dp
from . import ops
from ._lib import ATTN_PARAM_FIELDS
from .model import Output, ParamStore, Placeholder, Session, StepStats  # noqa: F401  (Session: re-exported for callers)

NAMES = ATTN_PARAM_FIELDS
# TF-1.1 variable names (original_attention.py:64-86; the cell is created under scope "s2vt" / "LSTM3", :109,129)
TF_NAMES = {n: n for n in NAMES}
TF_NAMES.update({"lstm3_W": "s2vt/LSTM3/basic_lstm_cell/weights", "lstm3_b": "s2vt/LSTM3/basic_lstm_cell/biases"})


def param_shapes(dim_image, n_words, dim_hidden):
    H, V, D = dim_hidden, n_words, dim_image
    return {"Wemb": (V, H), "encode_image_W": (D, H), "encode_image_b": (H,), "embed_att_w": (H, 1), "embed_att_Wa": (H, H),
            "embed_att_Ua": (H, H), "embed_att_ba": (H,), "embed_word_W": (H, V), "embed_word_b": (V,), "embed_nn_Wp": (3 * H, H),
            "embed_nn_bp": (H,), "lstm3_W": (3 * H, 4 * H), "lstm3_b": (4 * H,)}


class Attention_Caption_Generator:
    """original_attention.py's Video_Caption_Generator.  Same constructor (:55); `m` / `beta` are the script's module-level
    regulariser constants (:299-300), `device` / `seed` additions."""

    def __init__(self, dim_image, n_words, dim_hidden, batch_size, n_video_lstm_steps, n_caption_lstm_steps, drop_out_rate,
                 bias_init_vector=None, m=0.5, beta=10.0, device="cuda", seed=1234):
        self.dim_image, self.n_words, self.dim_hidden, self.batch_size = dim_image, n_words, dim_hidden, batch_size
        self.n_video_lstm_steps, self.n_caption_lstm_steps, self.drop_out_rate = n_video_lstm_steps, n_caption_lstm_steps, drop_out_rate
        self.m, self.beta = float(m), float(beta)
        self.device = torch.device(device)
        H = dim_hidden
        self.dims = ops.make_dims(dim_image, n_words, H, H, n_video_lstm_steps, n_caption_lstm_steps)
        self.store = ParamStore(param_shapes(dim_image, n_words, H), self.device, order=NAMES, tf_names=TF_NAMES,
                                params_factory=ops.make_attn_params)
        self.p, self.g = self.store.p, self.store.g
        # the reference initialisers (:65-86): U(-0.1, 0.1), zero biases; TF-default Glorot-uniform for the BasicLSTMCell kernel
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for n in NAMES:
            shp = self.store.shapes[n]
            if n.endswith("_b") or n in ("embed_att_ba", "embed_nn_bp"):
                self.p[n].zero_()
                continue
            a = 0.1 if n != "lstm3_W" else math.sqrt(6.0 / (shp[0] + shp[1]))
            self.p[n].copy_(((torch.rand(shp, generator=gen) * 2 - 1) * a).to(self.device))
        if bias_init_vector is not None:
            self.p["embed_word_b"].copy_(torch.as_tensor(np.asarray(bias_init_vector, np.float32)).to(self.device))
        self.global_step = 0
        self.adam_t = 0
        self.world_size, self.rank, self.dp_overlap = 1, 0, False        # data parallel: one all-reduce of the flat bucket (dist.py)
        self.dropout_seed = seed + 1
        self._gscale = torch.ones(1, dtype=torch.float32, device=self.device)
        self._applied = torch.zeros(1, dtype=torch.int32, device=self.device)   # step number of the last Adam update the device APPLIED
        self._row_ids_cache = {}

    # ------------------------------------------------------------------------------------------ utilities
    def _dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def _row_ids(self, B, video_base=0):
        key = (B, int(video_base))
        hit = self._row_ids_cache.get(key)
        if hit is None:
            hit = self._row_ids_cache[key] = ((torch.arange(B, dtype=torch.int32, device=self.device) + video_base).contiguous(),
                                              torch.zeros(B, dtype=torch.int32, device=self.device))
        return hit

    def load(self, arrays):
        """Variables by our names or by their TF checkpoint names."""
        inv = {v: k for k, v in TF_NAMES.items()}
        for k, v in arrays.items():
            k = inv.get(k, k)
            if k in self.p:
                self.p[k].copy_(torch.as_tensor(np.asarray(v, np.float32)).reshape(self.p[k].shape).to(self.device))

    def state_dict(self, with_optimizer=True):
        return self.store.state_dict(self.global_step if with_optimizer else None, self.adam_t, step_name="Variable")

    def load_state_dict(self, sd):
        loaded = self.store.load_state_dict(sd)
        if self.store.restored_step is not None:
            self.global_step = self.store.restored_step
            self.adam_t = self.store.restored_adam_t if self.store.restored_adam_t is not None else self.global_step
        return loaded

    # ------------------------------------------------------------------------------------------ forward graphs
    def forward(self, video, caption=None, greedy=False, keep=1.0, seed=0, video_base=0):
        """Teacher-forced logits [B,Tc,V] + alphas [Tc,Tv,B] (build_model, :95-147), or the greedy ids + alphas
        (build_sampler) when greedy=True.  LSTM3 output dropout as DropoutWrapper (:78) when keep < 1."""
        video = self._dev(video, torch.float32)
        B = video.shape[0]
        if greedy:
            ids, al = ops.attn_decode_greedy(self.dims, self.store.params, video, video_base, want_alphas=True)
            return None, al, ids
        cap = self._dev(caption, torch.int32)
        vid, sid = self._row_ids(B, video_base)
        logits, al, _ = ops.attn_teacher_forced_fwd(self.dims, self.store.params, video, cap, keep, seed, vid, sid, want_alphas=True)
        return logits.view(self.n_caption_lstm_steps, B, -1).transpose(0, 1), al, None

    @staticmethod
    def _active_steps(mask, Tc):
        """Leading decode steps at which any row is unmasked, read off a HOST mask (None for a device tensor: no sync)."""
        if isinstance(mask, torch.Tensor):
            if mask.is_cuda:
                return None
            mask = mask.numpy()
        m = np.asarray(mask)
        live = np.flatnonzero((m != 0).any(axis=0))
        return max(1, min(Tc, int(live[-1]) + 1)) if live.size else 1

    def _loss_forward(self, video, caption, caption_mask, keep, steps, video_base=0):
        """The unroll + softmax-NLL forward/backward; leaves d/dlogits and the activations ready for the backward."""
        video = self._dev(video, torch.float32)
        cap = self._dev(caption, torch.int32)
        mask = self._dev(caption_mask, torch.float32)
        B, Tc = cap.shape
        steps = Tc if steps is None else steps
        R = steps * B
        vid, sid = self._row_ids(B, video_base)
        seed = self.dropout_seed + 104729 * self.global_step
        logits, _, ws = ops.attn_teacher_forced_fwd(self.dims, self.store.params, video, cap, keep, seed, vid, sid, steps=steps)
        # time-major targets / coefficients (cross_entropy * caption_mask[:, i], :145), the regulariser's beta * mask (:123,144: identically
        # zero while Tv <= 8 -- the softmax sums to 1 > m) and sum(mask): one library launch
        want_reg = self.n_video_lstm_steps > 8 and self.beta != 0.0
        target, coef, reg, msum = ops.attn_loss_inputs(cap, mask, self.beta, want_reg)
        target, coef = target[:R], coef[:R]
        reg = reg[:R] if reg is not None else None
        nll, _ = ops.softmax_nll_fwd_bwd(logits, target, coef, 0.0)    # logits <- coef * (softmax - onehot)
        return dict(video=video, B=B, steps=steps, dlogits=logits, ws=ws, coef=coef, nll=nll, reg=reg, msum=msum, keep=keep, seed=seed,
                    vid=vid, sid=sid)

    def loss(self, video, caption, caption_mask, keep=None):
        """build_model's loss tensor (:149) evaluated (forward only)."""
        keep = self.drop_out_rate if keep is None else keep
        c = self._loss_forward(video, caption, caption_mask, keep, None)
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        ops.attn_step_scalars(self.dims, c["B"], c["ws"], c["coef"], c["nll"], c["reg"], self.m, c["msum"], c["msum"], out, None, None)
        return out[0]

    def xe_update(self, video, caption, caption_mask, lr, clip_norm=10.0, keep=None, active_steps="auto", video_base=0,
                  beta1=0.9, beta2=0.999, eps=1e-8):
        """One training step (:436-441): loss = (sum ce*mask + sum regulariser) / sum(mask), tf.gradients through the unroll,
        clip_by_global_norm(10), TF-form Adam.  active_steps "auto": behind the batch's longest caption every position is
        masked and adds exact zeros, so with a host-resident mask only the leading steps are unrolled (single process)."""
        keep = self.drop_out_rate if keep is None else keep
        Tc = self.n_caption_lstm_steps
        steps = Tc
        if active_steps == "auto":
            s = self._active_steps(caption_mask, Tc) if not dp.active() else None
            steps = Tc if s is None else s
        elif active_steps:
            steps = max(1, min(Tc, int(active_steps)))
        c = self._loss_forward(video, caption, caption_mask, keep, steps, video_base)
        st = self.store
        ops.zero_(st.grad)
        ops.attn_bptt_bwd(self.dims, st.params, st.grads, c["video"], c["dlogits"], c["ws"], c["steps"], c["reg"], self.m, keep, c["seed"],
                          c["vid"], c["sid"])
        gsum = dp.allreduce_bucket(st.grad, st.numel, c["msum"])                 # RCCL: one flat bucket + sum(mask) in its tail
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        sumsq = torch.empty(1, dtype=torch.float32, device=self.device)
        ops.attn_step_scalars(self.dims, c["B"], c["ws"], c["coef"], c["nll"], c["reg"], self.m, c["msum"], gsum, loss, self._gscale, sumsq)
        ops.grad_finalize(st.grad[:st.numel], st.theta, self._gscale, 0.0, sumsq)
        self.global_step += 1
        self.adam_t += 1
        ops.adam_tf(st.theta, st.grad[:st.numel], st.m, st.v, sumsq, clip_norm, lr, self.adam_t, beta1, beta2, eps, applied_step=self._applied)
        return StepStats(loss[0], sumsq, c["msum"][0])

    # ------------------------------------------------------------------------------------------ what the training drivers use
    def active_steps(self, mask):
        return self._active_steps(mask, self.n_caption_lstm_steps)

    def sample(self, video, K=0, with_greedy=True, seed=None, video_base=0):
        """(None, greedy ids [B, Tc]) -- the greedy sampler in the shape train_common.greedy_eval expects (this model has no
        multinomial sampler: original_attention.py trains with cross entropy only)."""
        assert K == 0 and with_greedy
        ids, _ = ops.attn_decode_greedy(self.dims, self.store.params, self._dev(video, torch.float32), video_base)
        return None, ids

    def set_step(self, global_step, adam_t=None):
        self.global_step = int(global_step)
        self.adam_t = int(global_step if adam_t is None else adam_t)
        self._applied.fill_(self.adam_t)

    def check_health(self):
        """Raise S2VTChainTimeout if a persistent recurrence (attn_chain.hip / attn_chain_bwd.hip) gave up a grid-wide wait since the
        last recover() -- a host-memory read, no synchronisation (as Video_Caption_Generator.check_health)."""
        if ops.chain_fault():
            from ._lib import S2VTChainTimeout
            raise S2VTChainTimeout("a persistent attention recurrence timed out (is another process running persistent kernels on this GPU?); "
                                   "the variables are intact: call recover() and repeat the step")

    def recover(self, disable_persistent=True):
        """After S2VTChainTimeout: synchronise, rewind the step counters to the last update the device applied, acknowledge the fault and
        switch to per-step launches (same bits).  Returns (step counter, updates skipped)."""
        torch.cuda.synchronize(self.device)
        applied = int(self._applied.item())
        ops.chain_ack(disable_persistent)
        lost = max(0, self.adam_t - applied)
        self.adam_t -= lost
        self.global_step -= lost
        return self.global_step, lost

    # ------------------------------------------------------------------------------------------ the reference surface
    def build_model(self):
        """(loss, video, caption, caption_mask) as original_attention.py:88-150."""
        B, Tc = self.batch_size, self.n_caption_lstm_steps
        video = Placeholder("video", (B, self.n_video_lstm_steps, self.dim_image), np.float32)
        caption = Placeholder("caption", (B, Tc), np.int32)
        caption_mask = Placeholder("caption_mask", (B, Tc), np.float32)

        def fn(v, c, m):
            return {"loss": float(self.loss(v, c, m))}
        return Output("loss", fn, [video, caption, caption_mask]), video, caption, caption_mask

    def build_generator(self):
        """(video, generated_words) as :155-199: greedy words [B, Tc] int64 for a batch_size block of videos."""
        video = Placeholder("video", (self.batch_size, self.n_video_lstm_steps, self.dim_image), np.float32)

        def fn(v):
            ids, _ = ops.attn_decode_greedy(self.dims, self.store.params, self._dev(v, torch.float32))
            return {"generated_words": ids.cpu().numpy().astype(np.int64)}
        return video, Output("generated_words", fn, [video])

    def build_sampler(self):
        """(sampled_captions, video, saved_alphas) as :201-251: dynamic batch; saved_alphas fetches as [Tc, Tv, B]
        (`n_caption_steps x n x b`, :251)."""
        video = Placeholder("video", (None, self.n_video_lstm_steps, self.dim_image), np.float32)

        def fn(v):
            ids, al = ops.attn_decode_greedy(self.dims, self.store.params, self._dev(v, torch.float32), want_alphas=True)
            return {"sampled_captions": ids.cpu().numpy().astype(np.int64), "saved_alphas": al.cpu().numpy()}
        return Output("sampled_captions", fn, [video]), video, Output("saved_alphas", fn, [video])

    # ---- the nodes train() adds around the model's graph (:430-441)
    def exponential_decay(self, start_learning_rate, decay_steps, decay_rate=0.5):
        """tf.train.exponential_decay(start_learning_rate, global_step, 10000, 0.5, staircase=True) (:431-432)."""
        def value():
            return float(start_learning_rate) * float(decay_rate) ** (self.global_step // int(decay_steps))
        out = Output("learning_rate", lambda: {"learning_rate": value()}, [])
        out.value = value
        return out

    def minimize(self, build_model_outputs, learning_rate, clip_norm=10.0):
        """train_op of :433-441: AdamOptimizer(learning_rate).compute_gradients(tf_loss) -> clip_by_global_norm(10) ->
        apply_gradients(global_step).  sess.run([train_op, tf_loss], feed) is ONE update; the loss fetched beside it is the
        one the update differentiated (pre-update weights, same dropout masks)."""
        loss, video, caption, caption_mask = build_model_outputs[:4]
        lr = learning_rate.value if hasattr(learning_rate, "value") else (lambda: float(learning_rate))

        def fn(v, c, m):
            st = self.xe_update(v, c, m, lr(), clip_norm=clip_norm)
            return {"train_op": None, "loss": float(st.loss)}
        return Output("train_op", fn, [video, caption, caption_mask], provides={loss: "loss"})
