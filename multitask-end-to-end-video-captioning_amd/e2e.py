"""CNN in the loop: the end-to-end scripts (e2e_tf_s2vt.py, reinforcement_e2e.py,
reinforce_multitask_e2e_attribute_loss.py -- BASELINE configs[4]).

The reference builds Inception-ResNet-v2 into every graph: frames [B, Tv, 299, 299, 3] -> base network with
is_training=False batch norm -> global average pool -> slim.dropout(keep 0.9; on in build_model / build_loss /
build_multinomial_sampler, off in build_sampler / build_generator) -> video [B, Tv, 1536] -> the captioner; one Adam
and one tf.clip_by_global_norm(., 10) span the CNN's and the captioner's variables (e2e_tf_s2vt.py:106-121,533-536;
reinforcement_e2e.py:958-971).

Here the captioner is the hand-written HIP path (model.Video_Caption_Generator) and the CNN is a torch module on
MIOpen (irv2.InceptionResnetV2 or any [n, 3, H, W] -> [n, dim_image] module); the seam between them is one tensor
each way:

    features = cnn(frames) --(detached, dropout applied)--> sample / teacher-forced forward / BPTT      [HIP]
    d_features = model.video_grad() * 1/sum(mask)   (s2vt_bptt_dvideo)  --> features.backward(d_features) [autograd]

The CNN's parameters and gradients live in two flat fp32 buffers (the modules hold views), so the data-parallel
all-reduce is one collective, the squared norm one dot product handed to the captioner's apply_gradients() as
`extra_sumsq` (so both halves are clipped by the SAME global norm), and the update one s2vt_adam_tf launch.
"""
from __future__ import annotations

import torch

from . import dist as dp
from . import ops


class EndToEnd:
    def __init__(self, model, cnn: torch.nn.Module, feature_keep: float | None = None, seed: int = 0, channels_last: bool = False,
                 miopen_find: bool | None = None):
        # miopen_find (env S2VT_MIOPEN_FIND=1): let MIOpen time its convolution algorithms per shape on first use
        # (torch.backends.cudnn.benchmark) instead of taking the immediate-mode heuristic -- fp32 either way; costs seconds at the first step
        import os
        if miopen_find is None:
            miopen_find = os.environ.get("S2VT_MIOPEN_FIND", "0") == "1"
        if miopen_find:
            torch.backends.cudnn.benchmark = True
        self.model, self.cnn = model, cnn.to(model.device)
        model.e2e = self                                     # the model's video placeholders take frames from here on (model.attach_cnn)
        self.channels_last = channels_last                   # feed the CNN NHWC activations (MIOpen picks its NHWC kernels)
        self.keep = model.dropout_rate if feature_keep is None else feature_keep      # slim.dropout(net, self.dropout_rate, ..)
        self.seed = seed
        params = [p for p in self.cnn.parameters() if p.requires_grad]
        n = sum(p.numel() for p in params)
        dev = model.device
        self.theta = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:                                     # parameters and their .grad become views of the flat buffers
            k = p.numel()
            self.theta[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.theta[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.params = params
        self._timing = None      # timing_enable(): [(start, end)] event pairs of the CNN forward / backward launches
        self.adam_t = 0          # Adam's count of updates applied to the CNN's moments: its own, not the captioner's -- a captioner restored
                                 # with its optimizer state (large t) beside fresh CNN moments would otherwise bias-correct zeros at step t

    # ---------------------------------------------------------------- where the step's time goes (bench.py: config.cnn_ms)
    def timing_enable(self, on: bool):
        """Bracket every CNN forward (extract) and every backward through the CNN with HIP events on the launching stream."""
        self._timing = {"fwd": [], "bwd": []} if on else None

    def timing_collect(self):
        """-> {"fwd": ms, "bwd": ms, "fwd_calls": n, "bwd_calls": n} summed over the brackets since timing_enable(True); synchronises."""
        t = self._timing or {"fwd": [], "bwd": []}
        torch.cuda.synchronize(self.model.device)
        out = {k: sum(a.elapsed_time(b) for a, b in t[k]) for k in ("fwd", "bwd")}
        out.update(fwd_calls=len(t["fwd"]), bwd_calls=len(t["bwd"]))
        if self._timing is not None:
            self._timing = {"fwd": [], "bwd": []}
        return out

    def _bracket(self, kind):
        class _B:
            def __enter__(b):
                if self._timing is not None:
                    b.a = torch.cuda.Event(enable_timing=True); b.a.record()

            def __exit__(b, *exc):
                if self._timing is not None:
                    e = torch.cuda.Event(enable_timing=True); e.record()
                    self._timing[kind].append((b.a, e))
        return _B()

    def conv_macs_per_frame(self, height, width):
        """Multiply-accumulates of one frame's forward through the CNN's convolution / linear layers (counted with forward hooks on a
        meta-device copy: no kernel runs) -- the figure bench.py prices the CNN half of the end-to-end step with."""
        import copy
        macs = [0]

        def hook(mod, inp, out):
            if isinstance(mod, torch.nn.Conv2d):
                macs[0] += out.numel() // out.shape[0] * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]
            elif isinstance(mod, torch.nn.Linear):
                macs[0] += mod.in_features * mod.out_features
        # (the live module's parameters are views of the flat buffer; a meta copy shares nothing with it)
        meta = copy.deepcopy(self.cnn).to("meta")
        hs = [m.register_forward_hook(hook) for m in meta.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
        with torch.no_grad():
            meta(torch.empty(1, 3, height, width, device="meta"))
        for h in hs:
            h.remove()
        return macs[0]

    # ---------------------------------------------------------------- features
    def _feature_dropout(self, f, B, Tv, video_base, draw):
        """slim.dropout(net, keep) on the pooled features [B*Tv, D] (e2e_tf_s2vt.py:120): (f / keep) * floor(keep + u), u from
        the library's counter-based dropout stream keyed by (GLOBAL video index, frame, unit) -- so n ranks x B/n videos
        apply the masks one rank x B would -- with one stream per step and per `draw` (the reference draws independent
        masks in every graph that has the op: build_multinomial_sampler and build_loss do not share one)."""
        dev = f.device
        vid = (torch.arange(B, dtype=torch.int32, device=dev) + int(video_base)).repeat_interleave(Tv).contiguous()
        frame = torch.arange(Tv, dtype=torch.int32, device=dev).repeat(B).contiguous()
        seed = self.seed + 15485863 * (self.model.global_step + 1)
        factor = ops.dropout_bwd(torch.ones_like(f), self.keep, seed, 768 + int(draw), vid, frame)     # = mask / keep
        return f * factor

    def extract(self, frames, dropout: bool, track: bool = False, video_base: int = 0, draws=(0,)):
        """frames [B, Tv, 3, H, W] fp32 in [-1, 1] -> (video [B, Tv, D] contiguous & detached, autograd handle or None).
        With several `draws` the CNN runs ONCE and one (video, handle) pair per independent dropout mask is returned; a draw of None is the
        pair without dropout (what the greedy graphs see: batch norm is in inference mode in every graph, so the pooled features are the same)."""
        B, Tv = frames.shape[:2]
        x = frames.to(self.model.device, torch.float32).reshape(B * Tv, *frames.shape[2:])
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        outs = []
        with torch.set_grad_enabled(track):
            with self._bracket("fwd"):
                raw = self.cnn(x)
            for d in draws:                                  # (a draw of None: the features WITHOUT dropout -- the greedy graphs', from the same CNN forward)
                f = self._feature_dropout(raw, B, Tv, video_base, d) if (dropout and d is not None and self.keep < 1.0) else raw
                f = f.reshape(B, Tv, -1)
                outs.append((f.detach().contiguous(), (f if track else None)))
        return outs[0] if len(outs) == 1 else outs

    # ---------------------------------------------------------------- CNN half of the update
    def _cnn_grads(self, handle, weight_decay, with_attr, handle_gt=None):
        """handle_gt: the ground-truth rows' own feature block of a mixed update (model.mixed_update(video_gt=)): video_grad() is then
        [(rep+1)*B, ...] -- the sampled blocks (summed into `handle`, they all read its block) and the ground-truth block."""
        def extra(gscale):
            m = self.model
            dv = m.video_grad() * gscale                       # d loss / d video, normalised by the GLOBAL sum(mask)
            dv_gt = None
            if handle_gt is not None:
                B = handle.shape[0]
                dv, dv_gt = dv[:-B].reshape(-1, B, *dv.shape[1:]).sum(0), dv[-B:]
            if with_attr and getattr(m, "_attr_ctx", None) is not None:
                dz, ascale = m._attr_ctx                        # attribute head: z = mean_t(video) @ attr_W + b
                WT = ops.transpose(m.store.p["attr_W"])
                dmean = ops.gemm([ops.operand(dz)], WT, None, M=dz.shape[0])
                dv += (dmean * (ascale / dv.shape[1]))[:, None, :]
            self.grad.zero_()
            with self._bracket("bwd"):
                if dv_gt is None:
                    handle.backward(dv)
                else:
                    torch.autograd.backward([handle, handle_gt], [dv, dv_gt])
            dp.allreduce_small(self.grad)
            if weight_decay:
                self.grad.add_(self.theta, alpha=weight_decay)
            return torch.dot(self.grad, self.grad).reshape(1)
        return extra

    def _cnn_decay_only(self, weight_decay):
        """The CNN half of fix_e2e_tf_s2vt.py's update: `net = tf.stop_gradient(net)` (:120) cuts the data gradient, but the
        weight-decay term sums l2_loss over ALL tf.trainable_variables() (:199, always-true predicate) and compute_gradients(tf_loss)
        (:534) covers them, so every CNN variable receives decay_value * theta, counts toward clip_by_global_norm(10) (:535) and is
        moved by Adam.  Identical on every rank: no all-reduce."""
        def extra(gscale):
            torch.mul(self.theta, float(weight_decay), out=self.grad)
            return torch.dot(self.grad, self.grad).reshape(1)
        return extra

    def _cnn_apply(self, lr, clip_norm):
        m = self.model
        self.adam_t += 1
        ops.adam_tf(self.theta, self.grad, self.m, self.v, m._sumsq, clip_norm, lr, self.adam_t)

    # ---------------------------------------------------------------- training steps
    def xe_step(self, frames, caption, caption_mask, lr, clip_norm=10.0, video_base=0, freeze_cnn=False, cnn_weight_decay=True,
                true_labels=None, attr_normalised=True):
        """One step of train() in e2e_tf_s2vt.py:482-700: label-smoothed XE through the CNN; weight decay on EVERY
        trainable variable (the always-true predicate at :199).
        freeze_cnn: the variant of fix_e2e_tf_s2vt.py (:120, :284: `net = tf.stop_gradient(net)`) -- the CNN runs in the loop,
        feature dropout and all, and no DATA gradient reaches it (no backward through the CNN).  As in that script its variables
        still receive the weight-decay gradient decay_value * theta (:199 sums over every trainable variable), are part of the
        joint clip norm and are moved by Adam (_cnn_decay_only).  cnn_weight_decay=False is the literal freeze instead -- a
        deviation from the reference, for callers that want the CNN untouched: only the captioner is clipped and updated.
        true_labels [B, label_dim]: the multitask cross-entropy objective of multitask_e2e_attribute_s2vt.py:208-222 (model.xe_update);
        the attribute head's gradient reaches the CNN through mean_t(video)."""
        multi = dict(true_labels=true_labels, attr_normalised=attr_normalised) if true_labels is not None else {}
        if freeze_cnn:
            assert not multi, "freeze_cnn is fix_e2e_tf_s2vt.py's variant: it has no attribute head"
            video, _ = self.extract(frames, dropout=True, track=False, video_base=video_base)
            if not cnn_weight_decay:
                return self.model.xe_update(video, caption, caption_mask, lr, clip_norm=clip_norm, video_base=video_base, decay_all=True)
            st = self.model.xe_update(video, caption, caption_mask, lr, clip_norm=clip_norm, video_base=video_base,
                                      extra_sumsq=self._cnn_decay_only(self.model.decay_value), decay_all=True)
            self._cnn_apply(lr, clip_norm)
            return st
        video, h = self.extract(frames, dropout=True, track=True, video_base=video_base)
        st = self.model.xe_update(video, caption, caption_mask, lr, clip_norm=clip_norm, video_base=video_base,
                                  extra_sumsq=self._cnn_grads(h, self.model.decay_value, bool(multi)), decay_all=True, **multi)
        self._cnn_apply(lr, clip_norm)
        return st

    def reinforce_update(self, frames, samples, mask, rewards, baseline, lr, clip_norm=10.0, video_base=0, true_labels=None, draw=1):
        """build_loss + train_op of the end-to-end REINFORCE scripts on captions ALREADY sampled (their train() runs the samplers
        in an earlier sess.run, reinforce_multitask_e2e_attribute_loss.py:1085-1087, then feeds the ids back with the same frames,
        :1113-1114): CNN forward with the loss graph's own slim.dropout mask (`draw`), model.reinforce_update on the pooled
        features, the gradient back through the CNN, one clip norm and one Adam over both halves."""
        video, h = self.extract(frames, dropout=True, track=True, video_base=video_base, draws=(draw,))
        st = self.model.reinforce_update(video, samples, mask, rewards, baseline, lr, clip_norm=clip_norm, video_base=video_base,
                                         true_labels=true_labels, extra_sumsq=self._cnn_grads(h, 0.0, true_labels is not None))
        self._cnn_apply(lr, clip_norm)
        return st

    def mixed_update(self, frames, samples, mask, rewards, baseline, gt_caption, gt_mask, lr, lambda_loss=0.5, clip_norm=5.0, video_base=0,
                     true_labels=None):
        """sum_loss = -(1 - lambda) * PG / norm + lambda * model_loss of reinforce_multitask_e2e_attribute_s2vt.py:850 through the CNN.
        The two graphs share the frames and the CNN (batch norm in inference mode, :126-130 and :302-306) and draw independent
        slim.dropout masks on the pooled features (:131, :307): ONE CNN forward, two feature blocks (draw 1: build_loss, draw 2:
        build_model), one teacher-forced pass of 2B distinct feature rows (model.mixed_update(video_gt=)), one backward through the
        CNN carrying both blocks' gradients.  model_loss decays EVERY trainable variable (:222), the CNN's included:
        lambda * decay_value * theta joins its gradient."""
        m = self.model
        (video_pg, h_pg), (video_xe, h_xe) = self.extract(frames, dropout=True, track=True, video_base=video_base, draws=(1, 2))
        lam = float(lambda_loss)
        st = m.mixed_update(video_pg, samples, mask, rewards, baseline, gt_caption, gt_mask, lr, lambda_loss=lam, clip_norm=clip_norm,
                            video_base=video_base, true_labels=true_labels, decay_all=True, video_gt=video_xe,
                            extra_sumsq=self._cnn_grads(h_pg, lam * m.decay_value, true_labels is not None, handle_gt=h_xe))
        self._cnn_apply(lr, clip_norm)
        return st

    def reinforce_step(self, frames, reward_fn, lr, K=1, clip_norm=10.0, video_base=0, true_labels=None, sample_seed=0):
        """One step of train() in reinforcement_e2e.py:1085-1140: sample (feature dropout ON, :399) and greedy
        (OFF, :466) through the CNN, reward_fn(samples[K*B,Tc], greedy[B,Tc]) -> (r[K*B], b[B]) on the host, then the
        REINFORCE update through the CNN (dropout ON, :308), clip 10 over all variables, one Adam."""
        m = self.model
        # ONE CNN forward for the step's three graphs (the reference runs the network in each of them; with batch norm in inference mode the
        # pooled features are the same everywhere and only slim.dropout differs): two independent feature-dropout masks -- the sampler's (:399)
        # and the loss graph's (:308) -- and the undropped features of the greedy graph (:466)
        (video_s, _), (video_u, h), (video_g, _) = self.extract(frames, dropout=True, track=True, video_base=video_base, draws=(0, 1, None))
        samples, _ = m.sample(video_s, K, False, seed=sample_seed, video_base=video_base)
        _, greedy = m.sample(video_g, 0, True, video_base=video_base)
        r, b = reward_fn(samples, greedy)
        r = torch.as_tensor(r, dtype=torch.float32)
        b = torch.as_tensor(b, dtype=torch.float32).repeat(K)
        st = m.reinforce_update(video_u, samples, None, r, b, lr, clip_norm=clip_norm, video_base=video_base,
                                true_labels=true_labels, extra_sumsq=self._cnn_grads(h, 0.0, true_labels is not None))
        self._cnn_apply(lr, clip_norm)
        st.samples, st.greedy = samples, greedy
        return st

    def evaluate_multilabel(self, frames, threshold=0.5):
        """evaluate_multilabel of reinforce_multitask_e2e_attribute_loss.py:606-626 through the CNN (batch norm and dropout in
        inference mode, :613-620): sigmoid(mean_t(video) . attr_W + attr_b) [B, label_dim] (device tensor).  `threshold` is accepted
        and unused, as there."""
        video, _ = self.extract(frames, dropout=False)
        return self.model.attribute_scores(video)

    def generate(self, frames, video_base=0):
        """build_generator / build_sampler through the CNN (dropout off): greedy ids [B, Tc]."""
        video, _ = self.extract(frames, dropout=False)
        return self.model.sample(video, 0, True, video_base=video_base)[1]
