"""Host-side mirror of the reference's model class and training steps.

``Video_Caption_Generator`` keeps the constructor and the ``build_*`` surface of the reference
(tf_s2vt.py:53-266, reinforcement_multisampling_tf_s2vt.py:63-466).  The reference's methods emit
a TF graph and return placeholders + output tensors that ``sess.run`` later evaluates; here they
return the same tuples of lightweight handles, and ``Session.run(fetches, feed_dict)`` evaluates
them by calling libs2vt_hip.so -- so a ``train()`` written against the reference keeps its shape
(see INTEGRATION.md).  All arithmetic is in the HIP library; torch supplies device memory,
streams and ``torch.distributed`` (RCCL).  There is no CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from . import dist as dp
from . import ops
from ._lib import PARAM_FIELDS

# TF-1.1 variable names of the reference checkpoints (optimistic_restore matches by name + shape,
# reinforcement_multisampling_tf_s2vt.py:47-61)
TF_NAMES = {
    "Wemb": "Wemb", "encode_image_W": "encode_image_W", "encode_image_b": "encode_image_b",
    "embed_word_W": "embed_word_W", "embed_word_b": "embed_word_b",
    "lstm1_W": "s2vt/LSTM1/basic_lstm_cell/weights", "lstm1_b": "s2vt/LSTM1/basic_lstm_cell/biases",
    "lstm2_W": "s2vt/LSTM2/basic_lstm_cell/weights", "lstm2_b": "s2vt/LSTM2/basic_lstm_cell/biases",
    "attr_W": "attr_W", "attr_b": "attr_b",
}
# flat layout: variables with weight decay first (tf_s2vt.py:163: every name without 'bias' --
# which keeps encode_image_b / embed_word_b IN), the two LSTM `biases` last
DECAYED = ("Wemb", "encode_image_W", "encode_image_b", "lstm1_W", "lstm2_W", "embed_word_W", "embed_word_b", "attr_W", "attr_b")
UNDECAYED = ("lstm1_b", "lstm2_b")


def param_shapes(dim_image, n_words, word_dim, lstm_dim, label_dim=0):
    E, H, V, D = word_dim, lstm_dim, n_words, dim_image
    s = {"Wemb": (V, E), "encode_image_W": (D, E), "encode_image_b": (E,), "lstm1_W": (E + H, 4 * H), "lstm1_b": (4 * H,),
         "lstm2_W": (2 * H + E, 4 * H), "lstm2_b": (4 * H,), "embed_word_W": (H, V), "embed_word_b": (V,)}
    if label_dim:
        s["attr_W"] = (D, label_dim)
        s["attr_b"] = (label_dim,)
    return s


class ParamStore:
    """All variables in ONE flat fp32 buffer (+ same-shaped grad / Adam m / Adam v buffers), each
    tensor a 256-byte aligned view: the optimizer and the RCCL all-reduce run over one range."""

    def __init__(self, shapes: dict, device, order=None, tf_names=None, params_factory=None):
        """order: the variables in bucket order (default: the S2VT layout -- weight-decayed variables first, the LSTM
        `biases` last); tf_names: variable -> TF checkpoint name; params_factory: dict of views -> the C struct of device
        pointers the library takes (default ops.make_params = s2vt_params)."""
        self.shapes = shapes
        self.tf_names = TF_NAMES if tf_names is None else tf_names
        if order is None:
            dec = [n for n in DECAYED if n in shapes]
            self.names = dec + [n for n in UNDECAYED if n in shapes]
        else:
            self.names = list(order)
        self.offsets = {}
        off = 0
        self.n_decayed = None
        for n in self.names:
            if order is None and n == UNDECAYED[0]:
                self.n_decayed = off
            self.offsets[n] = off
            off += (int(np.prod(shapes[n])) + 63) // 64 * 64
        if self.n_decayed is None:
            self.n_decayed = off
        self.numel = off
        # one spare slot block at the end of the gradient buffer carries sum(mask) through the all-reduce
        self.theta = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off + 64, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.p = {n: self._view(self.theta, n) for n in self.names}
        self.g = {n: self._view(self.grad, n) for n in self.names}
        on_gpu = torch.device(device).type == "cuda"      # (a CPU store serves checkpoint conversion / tests: no kernel can take it)
        factory = ops.make_params if params_factory is None else params_factory
        self.params = factory(self.p) if on_gpu else None
        self.grads = factory(self.g) if on_gpu else None

    def _view(self, flat, n):
        k = int(np.prod(self.shapes[n]))
        return flat[self.offsets[n]:self.offsets[n] + k].view(*self.shapes[n])

    def load(self, arrays: dict):
        for n, a in arrays.items():
            if n in self.p:
                self.p[n].copy_(torch.as_tensor(np.asarray(a)).to(self.p[n].device))

    def state_dict(self, global_step=None, adam_t=None, step_name="g_step", counter_dtype=np.int64):
        """Checkpoint with the reference's TF variable names.  With global_step given, also what a tf.train.Saver created
        AFTER the optimizer stores beside the variables (reinforcement_multisampling_tf_s2vt.py:661; the saver of
        tf_s2vt.py:440 is created before the optimizer and holds the model variables only): the Adam slots `<var>/Adam` (m),
        `<var>/Adam_1` (v), `beta1_power`, `beta2_power` and the step counter, so a resumed run continues the moments, the
        bias correction and the learning-rate staircase.  The counter is stored under `step_name` -- 'g_step' is the
        REINFORCE script's name (:637), 'Variable' what the unnamed tf.Variable(0, trainable=False) of the other scripts gets
        (tf_s2vt.py:441) -- and under 'global_step'.  adam_t: Adam's own count of applied updates (what beta*_power encode);
        defaults to global_step."""
        sd = {self.tf_names[n]: self.p[n].detach().cpu().numpy() for n in self.names}
        if global_step is not None:
            t = int(global_step if adam_t is None else adam_t)
            for n in self.names:
                sd[self.tf_names[n] + "/Adam"] = self._view(self.m, n).detach().cpu().numpy()
                sd[self.tf_names[n] + "/Adam_1"] = self._view(self.v, n).detach().cpu().numpy()
            sd["beta1_power"] = np.float32(0.9 ** (t + 1))      # TF keeps beta^(t+1) after t applied steps
            sd["beta2_power"] = np.float32(0.999 ** (t + 1))
            sd["global_step"] = np.int64(global_step)
            sd[step_name] = counter_dtype(global_step)       # (tf.Variable(0, trainable=False) is DT_INT32: the TF writers pass np.int32)
            sd["adam_t"] = np.int64(t)                       # Adam's update count, stated (beta1_power = 0.9^(t+1) goes denormal past t ~ 800)
        return sd

    def load_state_dict(self, sd: dict):
        """optimistic_restore: load every variable whose NAME and SHAPE match, ignore the rest (Adam slots included).
        Returns the loaded names; `self.restored_step` is the checkpoint's step counter or None (names 'global_step',
        'g_step' -- reinforcement_multisampling_tf_s2vt.py:637 -- or 'Variable', the unnamed counter of tf_s2vt.py:441);
        `self.restored_adam_t` is Adam's count of applied updates decoded from `beta1_power` (= 0.9^(t+1)) or None: the two
        differ when a REINFORCE run starts from an XE checkpoint (slots and beta powers match by name, 'g_step' does not)."""
        inv = {v: k for k, v in self.tf_names.items()}
        loaded = []
        self.restored_step = None
        self.restored_adam_t = None
        sd = dict(sd)
        if "_s2vt/adam_t" in sd:                                 # the TF-format files' private name for the same count
            sd.setdefault("adam_t", sd.pop("_s2vt/adam_t"))
        explicit_t = "adam_t" in sd and np.ndim(sd["adam_t"]) == 0
        if explicit_t:
            self.restored_adam_t = int(sd["adam_t"])

        def put(dst, arr):
            dst.copy_(torch.as_tensor(np.asarray(arr, dtype=np.float32)).to(dst.device))
        for name, arr in sd.items():
            base, slot = name, None
            for suffix in ("/Adam_1", "/Adam"):
                if name.endswith(suffix):
                    base, slot = name[:-len(suffix)], suffix
                    break
            n = inv.get(base, base)
            if n in self.p and tuple(np.shape(arr)) == tuple(self.shapes[n]):
                put(self.p[n] if slot is None else self._view(self.m if slot == "/Adam" else self.v, n), arr)
                loaded.append(name)
            elif name in ("global_step", "g_step", "Variable") and np.ndim(arr) == 0 and np.issubdtype(np.asarray(arr).dtype, np.integer):
                if name == "global_step" or self.restored_step is None:
                    self.restored_step = int(arr)
                loaded.append(name)
            elif name == "adam_t" and np.ndim(arr) == 0:
                self.restored_adam_t = int(arr)                  # stated outright: wins over the decoded beta1_power
                explicit_t = True
                loaded.append(name)
            elif name == "beta1_power" and np.ndim(arr) == 0:
                b = float(arr)
                if 0.0 < b < 1.0 and not explicit_t:
                    self.restored_adam_t = max(0, int(round(math.log(b) / math.log(0.9))) - 1)
                loaded.append(name)
        return loaded


def init_reference(store: ParamStore, seed: int = 1234):
    """The reference initialisers (tf_s2vt.py:69-84): U(-0.1, 0.1) for Wemb / encode_image_W /
    embed_word_W / attr_W, TF-default Glorot-uniform for the BasicLSTMCell kernels, zero biases."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for n in store.names:
        shp = store.shapes[n]
        if n.endswith("_b"):
            store.p[n].zero_()
            continue
        a = 0.1 if not n.startswith("lstm") else math.sqrt(6.0 / (shp[0] + shp[1]))
        store.p[n].copy_(((torch.rand(shp, generator=g) * 2 - 1) * a).to(store.p[n].device))


# ---------------------------------------------------------------------------------------------
# graph handles + session shim
# ---------------------------------------------------------------------------------------------
class Placeholder:
    def __init__(self, name, shape, dtype):
        self.name, self.shape, self.dtype = name, shape, dtype

    def __repr__(self):
        return f"<placeholder {self.name} {self.shape}>"


class Output:
    """A fetchable value: evaluated by Session.run from the fed placeholders.
    provides: {other Output: key} -- values of OTHER fetches this one's evaluation yields as a by-product (a train_op
    evaluates its loss on the way: sess.run([train_op, tf_loss]) is one pass, and the loss is the pre-update one, as in
    TF where train_op depends on the loss node)."""

    def __init__(self, name, fn, inputs, provides=None):
        self.name, self.fn, self.inputs = name, fn, inputs
        self.provides = provides or {}

    def __repr__(self):
        return f"<output {self.name}>"


class Session:
    """Plays sess.run(fetches, feed_dict): groups fetches by the graph call that produces them so one
    library call serves all outputs of the same build_* graph."""

    def __init__(self, model):
        self.model = model

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        feed = feed_dict or {}
        cache, given = {}, {}

        def evaluate(f):
            key = id(f.fn)
            if key not in cache:
                missing = [p.name for p in f.inputs if p not in feed]
                if missing:
                    raise KeyError(f"Session.run: fetch {f.name!r} needs a value for placeholder(s) {missing}")
                cache[key] = f.fn(*[feed[p] for p in f.inputs])
            return cache[key]
        # As in TF, every fetch of one run() sees the SAME pre-update state: (1) fetches that no state-changing fetch yields
        # (build_model's probs next to a train_op, a sampler's ids) are evaluated first, on the variables, step counter and
        # dropout seed the update is about to read; (2) then the state-changing fetches, whose by-products (the loss the
        # update differentiated) stand for the fetches they provide.
        will_give = {id(o) for f in fl if f.provides for o in f.provides}
        early = {}
        for f in fl:
            if not f.provides and id(f) not in will_give:
                early[id(f)] = evaluate(f)[f.name]
        for f in fl:
            if f.provides:
                res = evaluate(f)
                for other, k in f.provides.items():
                    given[id(other)] = res[k]
        out = [early[id(f)] if id(f) in early else (given[id(f)] if id(f) in given else evaluate(f)[f.name]) for f in fl]
        return out[0] if single else out


class LazyScalar:
    """A device-side scalar that is only computed when somebody reads it (float(x), x.item()): the multitask steps' attribute-loss term
    bce.sum() * scale is a reporting value -- two tensor-library launches per step when formed eagerly, none when nobody asks."""

    def __init__(self, fn, unscaled=None):
        self._fn, self._val, self._unscaled = fn, None, unscaled

    def tensor(self):
        if self._val is None:
            self._val = self._fn()
        return self._val

    def __float__(self):
        return float(self.tensor())

    def unscaled(self):
        """The term without the objective's weight on it (sum(bce) * normaliser, not alpha * that), when the maker provided it."""
        return self._unscaled() if self._unscaled is not None else self.tensor()

    def item(self):
        return float(self)


@dataclass
class StepStats:
    loss: torch.Tensor          # device scalar
    grad_sumsq: torch.Tensor    # device scalar (global norm squared, before clipping)
    mask_sum: torch.Tensor


class Video_Caption_Generator:
    """Same constructor as the reference, positional order included: the twelve arguments of tf_s2vt.py:54-66, then
    `width, height, channels, feature_dim, label_dim, alpha` of the multitask / end-to-end classes
    (reinforce_multitask_e2e_attribute_loss.py:71-76, reinforce_multitask_e2e_attribute_s2vt.py:71-76).  Defaults differ in
    ONE place, stated: label_dim defaults to 0 (= the base class of tf_s2vt.py, no attribute head; the multitask class says 400) --
    multitask.Video_Caption_Generator is this class with the multitask defaults (label_dim=400, alpha=0.2, multisample=1).
    `device`, `seed`, `multisample` (the K the reference hard-codes into build_loss, batch_size*8 at
    reinforcement_multisampling_tf_s2vt.py:228) are additions.  width / height / channels describe the frame placeholder the
    build_* graphs expose once a CNN is attached (attach_cnn); feature_dim is the attribute head's input width and must equal
    dim_image (it reads mean_t(video), :375)."""

    def __init__(self, dim_image, n_words, word_dim, lstm_dim, batch_size, n_lstm_steps, n_video_lstm_step,
                 n_caption_lstm_step, bias_init_vector=None, loss_weight=1, decay_value=0.00005, dropout_rate=0.9,
                 width=299, height=299, channels=3, feature_dim=None, label_dim=0, alpha=0.0, device="cuda", seed=1234,
                 multisample=8):
        self.dim_image, self.n_words, self.word_dim, self.lstm_dim = dim_image, n_words, word_dim, lstm_dim
        self.batch_size, self.n_lstm_steps = batch_size, n_lstm_steps
        self.n_video_lstm_step, self.n_caption_lstm_step = n_video_lstm_step, n_caption_lstm_step
        self.loss_weight, self.decay_value, self.dropout_rate = loss_weight, decay_value, dropout_rate
        self.width, self.height, self.channels = width, height, channels
        self.feature_dim = dim_image if feature_dim is None else feature_dim
        if label_dim and self.feature_dim != dim_image:
            raise ValueError(f"feature_dim ({self.feature_dim}) must equal dim_image ({dim_image}): the attribute head reads mean_t(video)")
        self.label_dim, self.alpha = label_dim, alpha
        self.e2e = None                          # e2e.EndToEnd once a CNN is attached: the video placeholders then take frames
        self.multisample = multisample          # the reference hard-codes batch_size*8 in build_loss (:228)
        self.device = torch.device(device)
        self.dims = ops.make_dims(dim_image, n_words, word_dim, lstm_dim, n_video_lstm_step, n_caption_lstm_step, label_dim)
        self.store = ParamStore(param_shapes(dim_image, n_words, word_dim, lstm_dim, label_dim), self.device)
        init_reference(self.store, seed)
        if bias_init_vector is not None:
            self.store.p["embed_word_b"].copy_(torch.as_tensor(np.asarray(bias_init_vector, np.float32)).to(self.device))
        self.global_step = 0                     # the step counter the learning-rate staircase and the noise seeds read
        self.adam_t = 0                          # Adam's count of applied updates (beta*_power); differs after a restore that
                                                 # matched the optimizer slots but not the counter (XE checkpoint -> REINFORCE run)
        self.sample_seed = seed
        self.dropout_seed = seed + 1
        self.world_size = 1
        self.rank = 0
        import os
        self.dp_overlap = os.environ.get("S2VT_DP_OVERLAP", "0") == "1"
        self._debug_checks = os.environ.get("S2VT_DEBUG_CHECKS", "0") == "1"
        self._applied = torch.zeros(1, dtype=torch.int32, device=self.device)   # step number of the last Adam update the device APPLIED
        self._row_id_cache = {}
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._gscale = torch.ones(1, dtype=torch.float32, device=self.device)
        self._ascale = torch.ones(1, dtype=torch.float32, device=self.device)

    # -------------------------------------------------------------------------------- utilities
    def _dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            # (a pinned host tensor is copied asynchronously: the caller keeps it unchanged until the copy has run -- the
            #  feature store's staging buffer is rewritten only after the step that used it has been synchronised)
            return a.to(device=self.device, dtype=dtype, non_blocking=(not a.is_cuda and a.is_pinned())).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()

    def _row_ids(self, B, rep, video_base):
        key = (B, rep, int(video_base))
        hit = self._row_id_cache.get(key)
        if hit is None:
            vid = (torch.arange(B, dtype=torch.int32, device=self.device) + video_base).repeat(rep)
            sid = torch.arange(rep, dtype=torch.int32, device=self.device).repeat_interleave(B)
            if len(self._row_id_cache) > 64:
                self._row_id_cache.clear()
            hit = self._row_id_cache[key] = (vid.contiguous(), sid.contiguous())
        return hit

    def _untile(self, v, N, host=None):
        """The reference feeds build_loss the feature block tiled K times, rows k*B+j = video j
        (reinforcement_multisampling_tf_s2vt.py:779-782).  Returns (the B distinct videos, B) when `v` is such a
        tiling -- LSTM1 and the frame embedding then run once per video; any other [N, ...] block is N videos.
        host: the feed as it arrived when it lives on the host (numpy / list, what Session.run receives): the K blocks are
        compared THERE, before the copy to the device, and a feed that is not a tiling takes the N-video path.  A device
        tensor is taken at the feed contract's word (comparing its copies would force a device-to-host synchronisation
        into every step; S2VT_DEBUG_CHECKS=1 verifies it)."""
        K = self.multisample
        if v.shape[0] == N and K > 1 and N % K == 0:
            B = N // K
            if host is not None:
                hb = np.asarray(host).reshape(K, B, -1)
                if not all(np.array_equal(hb[0], hb[k]) for k in range(1, K)):
                    return v.contiguous(), v.shape[0]
            blocks = v.view(K, B, *v.shape[1:])
            if self._debug_checks:
                assert bool((blocks == blocks[0]).all()), "build_loss feed: rows k*B+j must repeat video j (np.tile of the feature block)"
            return blocks[0].contiguous(), B
        return v.contiguous(), v.shape[0]

    # -------------------------------------------------------------------------------- video feeds (features, or frames once a CNN is attached)
    def attach_cnn(self, cnn, **kw):
        """Put a CNN in front of every graph, as the end-to-end scripts build Inception-ResNet-v2 into each build_*
        (e2e_tf_s2vt.py:106-121, reinforce_multitask_e2e_attribute_loss.py:116-131): from here on the video placeholders are
        the reference's frame placeholders [batch, n_video_lstm_step, height, width, channels]; build_sampler / build_generator /
        evaluate_multilabel run the CNN in inference mode, build_model / build_loss / build_multinomial_sampler add slim.dropout
        (keep = dropout_rate) on the pooled features, and every train op back-propagates into the CNN and clips / updates both
        halves together.  Returns the e2e.EndToEnd that owns the CNN's flat parameter, gradient and Adam buffers (kw: its
        constructor's).  A feature block [n, Tv, dim_image] fed to the same placeholder is still taken as features."""
        from .e2e import EndToEnd
        return EndToEnd(self, cnn, **kw)                         # (registers itself as self.e2e)

    def _video_placeholder(self, batch):
        if self.e2e is not None:
            return Placeholder("video_frames", (batch, self.n_video_lstm_step, self.height, self.width, self.channels), np.float32)
        return Placeholder("video", (batch, self.n_video_lstm_step, self.dim_image), np.float32)

    @staticmethod
    def _is_frames(v):
        return len(v.shape if hasattr(v, "shape") else np.shape(v)) == 5

    def _frames(self, v):
        """A frame feed as [B, Tv, channels, H, W]: the reference's placeholders are [B, Tv, H, W, channels]
        (reinforce_multitask_e2e_attribute_loss.py:118); a channel-first block (what e2e.EndToEnd takes) passes through."""
        if self.e2e is None:
            raise ValueError("a frame block [B, Tv, H, W, C] was fed but no CNN is attached (attach_cnn); feed features [B, Tv, dim_image]")
        f = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32))
        if f.shape[-1] != self.channels and f.shape[2] == self.channels:
            return f
        if f.shape[-1] != self.channels:
            raise ValueError(f"frame feed {tuple(f.shape)}: expected [B, Tv, H, W, {self.channels}]")
        return f.permute(0, 1, 4, 2, 3)

    def _features(self, v, dropout=False, draw=0, video_base=0):
        """A video-placeholder feed -> the feature block [n, Tv, dim_image] on the device (frames go through the attached CNN,
        batch norm in inference mode; dropout / draw: slim.dropout on the pooled features with the draw-th independent mask)."""
        if not self._is_frames(v):
            return self._dev(v, torch.float32)
        f = self._frames(v)                                          # (raises when no CNN is attached)
        return self.e2e.extract(f, dropout=dropout, track=False, video_base=video_base, draws=(draw,))[0]

    def _decay_everything(self):
        """Which weight-decay predicate the graphs use: tf_s2vt.py:163 skips names with 'bias'; the multitask and end-to-end
        classes' `if 'bias' or 'BatchNorm' not in v.name` is always true (reinforce_multitask_e2e_attribute_loss.py:222,
        e2e_tf_s2vt.py:199), so with an attribute head or an attached CNN every trainable variable is decayed.
        `self.decay_all_variables` (None = this rule) overrides."""
        forced = getattr(self, "decay_all_variables", None)
        return (self.label_dim > 0 or self.e2e is not None) if forced is None else bool(forced)

    def l2_term(self):
        """weight_decay_loss of the build_model graph as a host float: decay_value * sum(l2_loss(v)) over the variables the
        predicate keeps (with an attached CNN: its trainable variables too, e2e_tf_s2vt.py:199)."""
        st = self.store
        th = st.theta[:st.numel] if self._decay_everything() else st.theta[:st.n_decayed]      # (pad slots are zeros)
        sq = torch.dot(th, th)
        if self.e2e is not None and self._decay_everything():
            sq = sq + torch.dot(self.e2e.theta, self.e2e.theta)
        return float(sq) * (0.5 * self.decay_value)

    # -------------------------------------------------------------------------------- samplers
    def sample(self, video, K, with_greedy=True, seed=None, video_base=0, stop_at_eos=False):
        """K multinomial captions per video (+ the greedy caption): (sampled [K*B,Tc], greedy [B,Tc])
        int32 device tensors, sample-major rows.  One encode, no host round trip per step.
        stop_at_eos (opt-in): rows that have emitted <eos> leave the decode loop (ids behind it read 0; the reference keeps
        sampling them, :318-337, and masks them afterwards) -- the same update, a shorter loop on a trained policy."""
        video = self._dev(video, torch.float32)
        out = ops.sample(self.dims, self.store.params, video, K, self.sample_seed if seed is None else seed, video_base,
                         with_greedy, stop_at_eos)
        ws, rows, serial = ops.sample.last_state[0], ops.sample.last_state[1], ops.sample.last_state[4]
        self._sampler_state = (ws, rows, video, video._version, video.shape[0], self.global_step, serial)
        return out

    def build_sampler(self):
        """Greedy sampler (tf_s2vt.py:217-266): returns (sampled_captions, video)."""
        video = self._video_placeholder(None)

        def fn(v):
            _, g = self.sample(self._features(v), 0, True)      # (frames: CNN + no feature dropout, reinforcement_e2e.py:466)
            return {"sampled_captions": g.cpu().numpy().astype(np.int64)}
        return Output("sampled_captions", fn, [video]), video

    def build_multinomial_sampler(self):
        """One multinomial caption per video (reinforcement_multisampling_tf_s2vt.py:294-339).
        Every run draws from a fresh Philox stream (the TF op is stateful too)."""
        video = self._video_placeholder(self.batch_size)
        state = {"calls": 0}

        def fn(v):
            state["calls"] += 1                                  # (frames: slim.dropout is ON in this graph, reinforcement_e2e.py:399)
            s, _ = self.sample(self._features(v, dropout=True, draw=0), 1, False, seed=self.sample_seed + 7919 * state["calls"])
            return {"sampled_captions": s.cpu().numpy().astype(np.int64)}
        return Output("sampled_captions", fn, [video]), video

    def build_generator(self, beam_size=1, length_normalization_factor=0.5, unshifted_softmax=False):
        """B=1 greedy generator (tf_s2vt.py:169-214): (video, sentence, probs) where sentence is a
        list of Tc scalar fetches (the reference's `break` at :212 never fires).
        unshifted_softmax=True reproduces the reference's word choice to the letter: argmax of
        exp(l) / sum(exp(l)) computed in fp32 without a max shift (:208-209) -- NaN, hence <eos>, once a logit
        reaches 88.72; the default is argmax of the logits, which is what that expression means wherever it is finite."""
        video = self._video_placeholder(1)
        Tc = self.n_caption_lstm_step

        def fn(v):
            v = self._features(v)
            if beam_size > 1:                    # final_beam_search.py:226-294 (B = 1, TopN beams)
                from .beam_generator import BeamSearchGenerator
                sent, _, _ = BeamSearchGenerator(self, beam_size, length_normalization_factor).generate(v)
                ids = np.zeros(Tc, np.int64)
                ids[:len(sent)] = sent[:Tc]
            elif unshifted_softmax:
                from .beam_generator import BeamSearchGenerator
                ids = BeamSearchGenerator(self, 1).generate_unshifted_softmax(v)
            else:
                _, g = self.sample(v, 0, True)
                ids = g.cpu().numpy().astype(np.int64)[0]
            return {f"word_{t}": ids[t] for t in range(Tc)}
        sentence = [Output(f"word_{t}", fn, [video]) for t in range(Tc)]
        return video, sentence, []

    # -------------------------------------------------------------------------------- training graphs
    @staticmethod
    def active_steps(mask):
        """Number of leading decode steps at which ANY row of `mask` [N, Tc] is unmasked (>= 1), or None when the mask
        does not live on the host (a device tensor would have to be synchronised for the answer).  Steps behind it add
        exact zeros to the loss and to every gradient, so the update methods unroll only this many
        (ops.teacher_forced_fwd(steps=)).  Data parallel with Q1's batch-mean cross entropy: pass the GLOBAL batch's mask
        (a rank whose own captions are shorter still contributes to the mean at a step another rank keeps alive)."""
        if isinstance(mask, torch.Tensor):
            if mask.is_cuda:
                return None
            mask = mask.numpy()
        m = np.asarray(mask)
        if m.ndim != 2 or m.size == 0:
            return None
        live = np.flatnonzero(np.asarray(m != 0).any(axis=0))
        return int(live[-1]) + 1 if live.size else 1

    def live_rows(self, mask, steps):
        """int32 device tensor of the unmasked (step, row) pairs of the first `steps` decode steps -- time-major index t * N + n,
        ascending -- or None when the mask is not host-resident or so few positions are masked that packing them out would not
        pay (> 85 % live), or a row has an unmasked position behind a masked one.  The vocabulary-sized kernels of an update
        (logits, softmax, dWout, dO2) and LSTM2's weight- and input-gradient products run on these rows only; the others' loss
        terms and gradient contributions are exact zeros (ops.teacher_forced_fwd(live=))."""
        if isinstance(mask, torch.Tensor):
            if mask.is_cuda:
                return None
            mask = mask.numpy()
        if mask is None or ((self.dims.lstm_dim | self.dims.word_dim) & 3):      # (packed rows move as 16-byte pieces)
            return None
        m = np.asarray(mask)
        if m.ndim != 2 or m.size == 0:
            return None
        on = m[:, :steps] != 0
        if (on[:, 1:] & ~on[:, :-1]).any():                     # an unmasked position behind a masked one: the recurrence-side
            return None                                         # products need every row's live positions to be a PREFIX of its steps
        tm = np.ascontiguousarray(on.T).reshape(-1)
        live = np.flatnonzero(tm).astype(np.int32)
        if live.size == 0 or live.size > 0.85 * tm.size:
            return None
        return torch.as_tensor(live).to(self.device)

    def _steps(self, active_steps, mask, local_ok=True):
        """Resolve an update method's `active_steps` argument: an int (clamped to 1..Tc), None / 0 = the full unroll, "auto" =
        from a host-resident mask when the local mask decides it (local_ok: single process, or rows that do not couple
        across ranks)."""
        Tc = self.n_caption_lstm_step
        if active_steps is None or active_steps == 0:
            return Tc
        if isinstance(active_steps, str):
            if active_steps != "auto":
                raise ValueError("active_steps: an int, None or 'auto'")
            s = self.active_steps(mask) if (local_ok or self.world_size == 1) else None
            return Tc if s is None else max(1, min(Tc, s))
        return max(1, min(Tc, int(active_steps)))

    def _forward_loss(self, video, caption, coef_tm, smoothing, rep, video_base, keep, reuse_sampler_state=False, target_tm=None,
                      steps=None, live=None, row_ids=None):
        """Teacher-forced forward + softmax-NLL fwd/bwd.  caption [N,Tc] int32 device, coef_tm
        time-major [Tc*N].  Leaves dlogits + activations ready for backward().  steps < Tc: only the first `steps` decode
        steps are unrolled (the caller vouches that coef_tm is zero behind them); nll / lp then have steps*N entries."""
        B = video.shape[0]
        N = caption.shape[0]
        steps = self.n_caption_lstm_step if steps is None else int(steps)
        R = steps * N
        # row_ids: the (video id, sample id) keys of the rows' dropout streams when they are not the sample-major numbering of
        # `video`'s own rows (mixed_update with a feature block per row block)
        vid, sid = self._row_ids(B, rep, video_base) if row_ids is None else row_ids
        seed = self.dropout_seed + 104729 * self.global_step
        state = None
        if reuse_sampler_state:
            ls = getattr(self, "_sampler_state", None)
            # (workspace, rows, the video tensor itself, its version counter, B, weights version): the tensor is held,
            # so its address cannot be recycled; an in-place write to it or an update of the variables in between
            # makes the saved LSTM1 trajectory stale
            ok = (ls is not None and ls[2] is video and ls[3] == video._version and ls[4] == B and ls[5] == self.global_step
                  and ops.sample.last_state is not None and ops.sample.last_state[4] == ls[6])   # nobody sampled into the shared workspace since
            if not ok:
                raise RuntimeError("reuse_sampler_state needs model.sample() on this very video tensor, with the current "
                                   "weights, as the last sampler call before the update")
            state = (ls[0], ls[1])
        logits, ws = ops.teacher_forced_fwd(self.dims, self.store.params, video, caption, N, keep, seed, vid, sid,
                                            sampler_state=state, steps=steps, live=live)
        if callable(coef_tm):            # host work (the reward) runs here, beside the forward just queued on the GPU
            coef_tm = coef_tm()
        target = caption.t().contiguous().view(-1) if target_tm is None else target_tm
        if isinstance(smoothing, torch.Tensor):
            smoothing = smoothing[:R]
        target, coef = target[:R], coef_tm[:R]                  # (time-major: a prefix)
        if live is not None:                                    # ... and of that, the unmasked positions (nll / lp: one entry per live row)
            ix = live.long()
            target, coef = target[ix].contiguous(), coef[ix].contiguous()
            if isinstance(smoothing, torch.Tensor):
                smoothing = smoothing[ix].contiguous()
        nll, lp = ops.softmax_nll_fwd_bwd(logits, target, coef, smoothing)
        self._coef_used = coef
        self._ctx = (video, N, logits, ws, keep, seed, vid, sid, steps, live)
        return nll, lp

    def backward(self, accumulate=False, overlap=None, keep_tail=False):
        """BPTT into the flat gradient bucket (accumulate=True: on top of what a previous pass left there).  Data parallel,
        overlap=True: the bucket is laid out [... lstm1_W | lstm2_W | embed_word_W | embed_word_b | ...]; the
        vocab-projection gradients are final after phase 1 and LSTM2's after phase 3, so their all-reduces are started
        there and run over xGMI beside the rest of the backward; apply_gradients() reduces what is left and waits for
        all of them.  overlap=None takes the model default `self.dp_overlap` (env S2VT_DP_OVERLAP, default OFF = one
        all-reduce of the whole bucket after the backward: the overlapped form has only been run on hardware with one
        rank under RCCL, tests/test_gpu_rccl.py -- no multi-GPU box was available to the build)."""
        if overlap is None:
            overlap = self.dp_overlap
        video, N, dlogits, ws, keep, seed, vid, sid, steps, live = self._ctx
        st = self.store
        if not accumulate:
            ops.zero_(st.grad[:st.numel] if keep_tail else st.grad)    # keep_tail: sum(mask) already sits in the tail slot (ops.caption_mask)
        self._pending = []
        self._early = None
        if dp.active() and overlap:
            def span(first, last):
                return st.offsets[first], st.offsets[last] + (int(np.prod(st.shapes[last])) + 63) // 64 * 64
            args = (self.dims, st.params, st.grads, video, N, dlogits, ws, keep, seed, vid, sid)
            ops.bptt_bwd(*args, phase=1, steps=steps, live=live)
            lo1, hi1 = span("embed_word_W", "embed_word_b")
            self._pending.append(dp.allreduce_async(st.grad[lo1:hi1]))
            # From here RCCL's kernels run on the communicator's stream beside ours.  A persistent recurrence needs ~every CU
            # co-resident (csrc/chain.hip) and must not be started into a chip that is partly taken: while a slice is in
            # flight the two backward recurrences take their per-step form (same bits; ops.chain_hold).
            with ops.chain_hold():
                ops.bptt_bwd(*args, phase=3, steps=steps, live=live)
                lo2, hi2 = span("lstm2_W", "lstm2_W")
                assert hi2 == lo1, "bucket layout: lstm2_W sits right below embed_word_W"
                self._pending.append(dp.allreduce_async(st.grad[lo2:hi2]))
                self._early = (lo2, hi1)                       # [lo2, hi1) is already on its way
                ops.bptt_bwd(*args, phase=4, steps=steps, live=live)
        else:
            ops.bptt_bwd(self.dims, st.params, st.grads, video, N, dlogits, ws, keep, seed, vid, sid, steps=steps, live=live)

    def video_grad(self):
        """d(loss * sum(mask)) / d(video) [B, Tv, dim_image] of the pass backward() just ran -- the gradient the
        end-to-end scripts push into the CNN (e2e_tf_s2vt.py:106-121).  Unnormalised, like the bucket before
        apply_gradients()."""
        video, N, _, ws, *_ = self._ctx
        return ops.bptt_dvideo(self.dims, self.store.params, video.shape[0], N, ws)

    def apply_gradients(self, mask_sum, lr, clip_norm, weight_decay=0.0, attr_scale=None, extra_sumsq=None, decay_all=False, loss_terms=None):
        """All-reduce (RCCL, one flat bucket + sum(mask) in its tail), 1/sum(mask), weight decay,
        tf.clip_by_global_norm, TF-form Adam (reinforcement_multisampling_tf_s2vt.py:643-652).
        attr_scale: constant normaliser of the attribute-head gradients (their range of the bucket is
        touched by the multilabel loss only, so it is finalised with its own scale).
        extra_sumsq: callable(gscale) -> device scalar, the squared norm of gradients held OUTSIDE the bucket (the CNN of
        the end-to-end scripts) that tf.clip_by_global_norm sees in the same list; it is called once 1/sum(mask) is
        known.  decay_all: the e2e scripts' always-true weight-decay predicate (e2e_tf_s2vt.py:199) -- the LSTM biases
        are decayed as well."""
        st = self.store
        early = getattr(self, "_early", None)
        if early is not None and dp.active():
            lo, hi = early
            if mask_sum is not None:
                st.grad[st.numel] = mask_sum
            pend = getattr(self, "_pending", [])
            pend.append(dp.allreduce_async(st.grad[:lo]))
            pend.append(dp.allreduce_async(st.grad[hi:]))            # includes the tail slot carrying sum(mask)
            dp.wait_all(pend, st.grad)
            self._pending, self._early = [], None
            gsum = st.grad[st.numel:st.numel + 1]
        else:
            gsum = dp.allreduce_bucket(st.grad, st.numel, mask_sum)      # mask_sum None: the tail slot already holds the local sum(mask)
        # one launch: 1 / global sum(mask), a fresh zeroed ||g||^2 accumulator, and the step's loss (loss_terms = (coef, nll,
        # local sum(mask)): sum(coef * nll) / sum(mask) of THIS rank's rows)
        self._sumsq = torch.empty(1, dtype=torch.float32, device=self.device)
        self._loss = torch.empty(1, dtype=torch.float32, device=self.device) if loss_terms is not None else None
        coef_, nll_, msum_ = loss_terms if loss_terms is not None else (None, None, None)
        ops.step_scalars(coef_, nll_, msum_, gsum, self._loss, self._gscale, self._sumsq)
        nd = st.n_decayed
        a0 = st.offsets.get("attr_W", nd)
        ops.grad_finalize(st.grad[:a0], st.theta[:a0], self._gscale, weight_decay, self._sumsq)
        if a0 < nd:
            want = 1.0 if attr_scale is None else float(attr_scale)
            if getattr(self, "_ascale_val", None) != want:          # (a constant of the run: one fill, not one per step)
                self._ascale.fill_(want)
                self._ascale_val = want
            ops.grad_finalize(st.grad[a0:nd], st.theta[a0:nd], self._ascale, weight_decay, self._sumsq)
        ops.grad_finalize(st.grad[nd:st.numel], st.theta[nd:], self._gscale, weight_decay if decay_all else 0.0, self._sumsq)
        if extra_sumsq is not None:
            self._sumsq += extra_sumsq(self._gscale)
        self.global_step += 1
        self.adam_t += 1
        self._sampler_state = None                 # the variables change: a saved sampler trajectory is stale from here on
        ops.adam_tf(st.theta, st.grad[:st.numel], st.m, st.v, self._sumsq, clip_norm, lr, self.adam_t, applied_step=self._applied)

    def set_step(self, global_step: int, adam_t=None):
        """Position the step counters (a restored checkpoint): the learning-rate / noise-seed counter and Adam's count of
        applied updates (defaults to the same number)."""
        self.global_step = int(global_step)
        self.adam_t = int(global_step if adam_t is None else adam_t)
        self._applied.fill_(self.adam_t)
        self._sampler_state = None

    # -------------------------------------------------------------------------------- persistent-recurrence health
    def check_health(self):
        """Raise S2VTChainTimeout if a persistent recurrence (csrc/chain.hip) gave up a grid-wide wait since the last
        recover().  A host-memory read, no device synchronisation: call it wherever the host has just synchronised anyway
        (after reading the loss / the ids) -- the training drivers do, every step.  Independently of this call, every
        library entry point that launches a recurrence or updates variables refuses to run while the fault is pending, and
        Adam launches queued behind the faulting kernel skip their update ON THE DEVICE: the variables are never touched by
        gradients of a starved recurrence."""
        if ops.chain_fault():
            from ._lib import S2VTChainTimeout
            raise S2VTChainTimeout("a persistent LSTM recurrence timed out (is another process running persistent kernels on this GPU?); "
                                   "the variables are intact: call recover() and repeat the step")

    def state_dict(self, with_optimizer=True, step_name="g_step"):
        """Variables under the reference's TF names (+ Adam slots, beta powers, the step counter with with_optimizer): store.state_dict with this
        model's counters -- the same two methods attention.Attention_Caption_Generator has."""
        return self.store.state_dict(self.global_step if with_optimizer else None, self.adam_t, step_name=step_name)

    def load_state_dict(self, sd):
        """store.load_state_dict + the counters, positioned as optimistic_restore does (set_step: the device-side count of APPLIED updates
        recover() rewinds to follows, and a sampler trajectory saved under the old variables is dropped).  A checkpoint that carries Adam's
        count (beta1_power / adam_t) but no step counter -- an XE checkpoint's optimizer slots under a REINFORCE run's names -- moves adam_t only."""
        loaded = self.store.load_state_dict(sd)
        step, t = self.store.restored_step, self.store.restored_adam_t
        if step is not None:
            self.set_step(step, t if t is not None else step)
        elif t is not None:
            self.set_step(self.global_step, t)
        self._sampler_state = None                   # the variables changed, whatever the counters say
        return loaded

    def recover(self, disable_persistent=True):
        """After S2VTChainTimeout: synchronise, learn from the device which update was the last one APPLIED (updates behind
        the fault were skipped), rewind the host-side step counter to it, acknowledge the fault and (default) switch the
        recurrences to per-step launches for the rest of the process (same bits, ~5 % slower).  Returns (the step counter
        the variables are at, the number of updates that were skipped); the caller repeats its work from there."""
        torch.cuda.synchronize(self.device)
        applied = int(self._applied.item())
        ops.chain_ack(disable_persistent)
        lost = max(0, self.adam_t - applied)
        self.adam_t -= lost
        self.global_step -= lost
        self._sampler_state = None
        self._pending, self._early = [], None
        return self.global_step, lost

    def reinforce_update(self, video, sampled, mask, rewards, baseline, lr, clip_norm=5.0, video_base=0, keep=None,
                         true_labels=None, reuse_sampler_state=False, extra_sumsq=None, reward_fn=None, active_steps="auto",
                         live_mask="auto"):
        """build_loss + the REINFORCE objective and train_op of train()
        (reinforcement_multisampling_tf_s2vt.py:227-292, 633-652): sampled [N,Tc] ids, mask [N,Tc] (None: derived from the
        ids on the device -- 1 up to and including the first <eos> -- with the coefficients, sum(mask) and the loss in three
        small library launches instead of ~20 tensor-library ones), rewards / baseline [N], rows sample-major over the B videos.
        reuse_sampler_state: the sample() call that produced `sampled` ran just before on this same video tensor
        with the current weights -- its LSTM1 trajectory is reused (the reference recomputes the whole unroll in
        build_loss).  With true_labels [B, label_dim] the multitask objective of
        reinforce_multitask_e2e_attribute_loss.py:957 is used instead:
            -(1-alpha) * PG / sum(mask) + alpha * sum(bce) / (label_dim * B).
        reward_fn: callable() -> (rewards [N], baseline [N]) evaluated on the host AFTER the teacher-forced forward has
        been queued on the GPU (which does not need them), so a host-side scorer (CIDEr-D) runs under it; `rewards` /
        `baseline` are ignored then.
        active_steps: decode steps to unroll -- "auto" (default): behind the longest sample of the batch every position is
        masked, so when `mask` is host-resident (numpy / CPU tensor, as the reference's loop has it) only the steps up
        to it run; an int: the caller's own count; None: all Tc.  Exact: the skipped steps add zeros.
        live_mask: a HOST mask [N, Tc] naming the unmasked positions ("auto": `mask` itself when it is host-resident; None:
        off): inside the unrolled steps the vocabulary projection, the softmax and their two gradient products run on those
        (step, row) pairs only (live_rows()) -- exact as well: a masked position's coefficient is zero."""
        steps = self._steps(active_steps, mask)
        live = self.live_rows(mask if isinstance(live_mask, str) else live_mask, steps) if live_mask is not None else None
        video = self._dev(video, torch.float32)
        cap = self._dev(sampled, torch.int32)
        st_ = self.store
        fused = mask is None        # the library derives the mask from the ids (1 up to and incl. the first <eos>, cider_evaluation.py:145-172)
        if fused:
            msum = torch.empty(1, dtype=torch.float32, device=self.device)
            mask, target_tm, _ = ops.caption_mask(cap, mask_sum=msum, mask_sum_copy=st_.grad[st_.numel:st_.numel + 1])
        else:
            mask = self._dev(mask, torch.float32)
            target_tm = None
        rep = cap.shape[0] // video.shape[0]
        multitask = true_labels is not None and self.label_dim > 0
        pg_w = (1.0 - self.alpha) if multitask else 1.0
        made = {}

        def make_coef():
            r, b = reward_fn() if reward_fn is not None else (rewards, baseline)
            r, b = self._dev(r, torch.float32), self._dev(b, torch.float32)
            if fused:
                made["coef"] = ops.pg_coef(mask, r, b, pg_w)
            else:
                made["coef"] = (mask * ((r - b) * pg_w)[:, None]).t().contiguous().view(-1)
            return made["coef"]
        keep = self.dropout_rate if keep is None else keep
        nll, _ = self._forward_loss(video, cap, make_coef, 0.0, rep, video_base, keep, reuse_sampler_state, target_tm=target_tm,
                                    steps=steps, live=live)
        coef = self._coef_used
        if not fused:
            msum = mask.sum().reshape(1)
        self.backward(keep_tail=fused)
        attr_scale = None
        attr_loss = None
        if multitask:
            attr_scale, attr_loss = self._attr_terms(video, true_labels)
        self.apply_gradients(None if fused else msum, lr, clip_norm, attr_scale=attr_scale, extra_sumsq=extra_sumsq, loss_terms=(coef, nll, msum))
        st = StepStats(self._loss[0], self._sumsq, msum[0])
        st.attr_loss = attr_loss
        st.mask = mask
        return st

    def xe_update(self, video, caption, caption_mask, lr, clip_norm=10.0, q1=True, smoothing=0.05, video_base=0, keep=None,
                  extra_sumsq=None, decay_all=False, active_steps="auto", live_mask="auto", true_labels=None, attr_normalised=True):
        """build_model + train_op of tf_s2vt.py:90-167,445-448 (label smoothing 0.05, Q1 batch-mean
        semantics, weight decay on the non-'bias' variables, clip 10).
        true_labels [B, label_dim] (model built with label_dim > 0): the multitask cross-entropy objective
            (1 - alpha) * XE / sum(mask) + weight decay + alpha * multilabel_loss
        of multitask_e2e_attribute_s2vt.py:208-222 (multilabel_loss = sum(bce) / (label_dim * B), attr_normalised=True) and of
        reinforce_multitask_e2e_attribute_loss.py:211-225 (multilabel_loss = sum(bce), attr_normalised=False).
        active_steps: as reinforce_update -- the padding behind the longest caption of the batch is not unrolled.  "auto"
        reads a host-resident caption_mask; data parallel with q1 it needs the GLOBAL batch's longest caption, which only the
        caller knows (train_xe passes it), so "auto" keeps the full unroll there.
        live_mask: as reinforce_update, without Q1 only (with Q1 every row of a live step carries the step's batch mean)."""
        steps = self._steps(active_steps, caption_mask, local_ok=not q1)
        live = None
        if not q1 and live_mask is not None:
            live = self.live_rows(caption_mask if isinstance(live_mask, str) else live_mask, steps)
        video = self._dev(video, torch.float32)
        cap = self._dev(caption, torch.int32)
        mask = self._dev(caption_mask, torch.float32)
        N = cap.shape[0]
        n_glob = float(N * self.world_size)
        keep = self.dropout_rate if keep is None else keep
        multitask = true_labels is not None and self.label_dim > 0
        lw = float(self.loss_weight) * (1.0 - float(self.alpha)) if multitask else self.loss_weight
        if not dp.active() and cap.shape[1] <= 128:
            # one process: coefficients, time-major targets and sum(mask) in ONE library launch (the expressions below, same order)
            coef, target_tm, msum = ops.xe_prep(mask.contiguous(), cap.contiguous(), lw, n_glob, q1)
            nll, _ = self._forward_loss(video, cap, coef, smoothing, 1, video_base, keep, steps=steps, live=live, target_tm=target_tm)
        else:
            colsum = mask.sum(0)
            if q1:
                dp.allreduce_small(colsum)
                coef = (colsum[:, None] / n_glob).expand(-1, N) * lw
            else:
                coef = mask.t() * lw
            coef = coef.contiguous().view(-1)
            nll, _ = self._forward_loss(video, cap, coef, smoothing, 1, video_base, keep, steps=steps, live=live)
            msum = mask.sum().reshape(1)
        coef = self._coef_used
        self.backward()
        attr_scale = attr_loss = None
        if multitask:
            attr_scale, attr_loss = self._attr_terms(video, true_labels, attr_normalised)
        self.apply_gradients(msum, lr, clip_norm, weight_decay=self.decay_value, extra_sumsq=extra_sumsq, decay_all=decay_all,
                             loss_terms=(coef, nll, msum), attr_scale=attr_scale)
        st = StepStats(self._loss[0], self._sumsq, msum[0])
        st.attr_loss = attr_loss
        return st

    def _attr_terms(self, video, true_labels, normalised=True):
        """The attribute head's half of a multitask update: its gradients into the bucket's attr range (unscaled), the scale
        alpha / (label_dim * B_global) (or alpha, for the un-normalised sum of reinforce_multitask_e2e_attribute_loss.py:221) that
        apply_gradients() finalises that range with, and the loss term as a lazily evaluated device scalar."""
        y = self._dev(true_labels, torch.float32)
        mean, z, bce = ops.attr_head_fwd(video, self.store.p["attr_W"], self.store.p["attr_b"], y)
        dz = ops.attr_head_bwd(mean, z, y, 1.0, self.store.g["attr_W"], self.store.g["attr_b"])
        scale = float(self.alpha) / float(self.label_dim * video.shape[0] * self.world_size) if normalised else float(self.alpha)
        self._attr_ctx = (dz, scale)                            # for callers that differentiate through `video` (e2e.py)
        norm = scale / float(self.alpha) if self.alpha else (1.0 / float(self.label_dim * video.shape[0] * self.world_size) if normalised else 1.0)
        return scale, LazyScalar(lambda bce=bce, sc=scale: bce.sum() * sc, unscaled=lambda bce=bce, nm=norm: bce.sum() * nm)

    def mixed_update(self, video, sampled, mask, rewards, baseline, gt_caption, gt_mask, lr, lambda_loss=0.5, clip_norm=5.0,
                     video_base=0, keep=None, q1=True, smoothing=0.05, true_labels=None, active_steps="auto", decay_all=False,
                     reuse_sampler_state=False, video_gt=None, extra_sumsq=None):
        """The mixed objective of reinforce_multitask_e2e_attribute_s2vt.py:850 (BASELINE configs[3]):
            sum_loss = -(1 - lambda) * PG / sum(mask_pg)  +  lambda * model_loss
        with PG the reward-scaled log-likelihood of the SAMPLED captions (build_loss) and model_loss the
        cross-entropy loss of build_model on the GROUND-TRUTH captions of the same videos (label smoothing, Q1,
        weight decay), clip 5, Adam.  The reference evaluates two graphs on the same videos; here the rep*B sampled rows and
        the B ground-truth rows are ONE teacher-forced pass of (rep+1)*B sample-major rows (the ground truth is "sample"
        number rep: its own dropout masks, label smoothing per row, each block's coefficient already divided by its GLOBAL
        mask sum), so the unrolls, the vocabulary products and the backward run once -- at B = 32, K = 1 that is 64 rows
        for the price of 32.  With true_labels [B, label_dim] (and a model built with label_dim > 0) the attribute head's
        term of reinforce_multitask_e2e_attribute_loss.py:957 is added: + alpha * sum(bce) / (label_dim * B_global) --
        the per-GPU shape of BASELINE configs[3] (attribute FC + XE mix + REINFORCE, K = 1).
        decay_all: SURVEY Q3, second half -- the weight-decay predicate of the multitask / e2e scripts,
        `if 'bias' or 'BatchNorm' not in v.name` (reinforce_multitask_e2e_attribute_s2vt.py:222), is always true, so model_loss
        decays EVERY variable, the LSTM `biases` included: their gradients carry lambda * decay_value * b.  False keeps
        tf_s2vt.py:163's filter (names without 'bias').
        reuse_sampler_state: as reinforce_update -- the sample() call that produced `sampled` ran just before on this very video tensor with the
        current weights; LSTM1 never sees a word or a dropout mask, so its trajectory is the same for the sampled AND the ground-truth rows of
        the pass and is taken from the sampler's workspace instead of being recomputed (one 25-step recurrence less per step).
        video_gt [B, Tv, dim_image]: the feature block the GROUND-TRUTH rows read when it is not `video` -- with the CNN in the loop the
        two graphs draw independent slim.dropout masks on the pooled features (reinforce_multitask_e2e_attribute_s2vt.py:131 in
        build_model, :307 in build_loss), so the same frames give each objective its own block.  The pass then runs on (rep+1)*B
        distinct feature rows (LSTM1 per row block) with the SAME dropout-stream keys as the shared-block form: row k*B+j keeps
        (video j, sample k).  video_grad() afterwards is [(rep+1)*B, Tv, dim_image], the sampled blocks first."""
        if active_steps == "auto":          # both blocks decide: the longest sample and the longest ground-truth caption
            sa, sb = self.active_steps(mask), self.active_steps(gt_mask)
            active_steps = None if (sa is None or sb is None or (q1 and self.world_size > 1)) else max(sa, sb)
        steps = self._steps(active_steps, None)
        live = None
        if not isinstance(mask, torch.Tensor) and not isinstance(gt_mask, torch.Tensor) and mask is not None and gt_mask is not None \
                and not (q1 and self.world_size > 1):
            # host masks: the sampled rows are live where their own mask is, the ground-truth rows wherever their coefficient is
            # non-zero -- with Q1 every row of a step at which ANY ground-truth caption is unmasked
            mp, mg = np.asarray(mask, np.float32), np.asarray(gt_mask, np.float32)
            if q1:
                mg = np.broadcast_to((mg != 0).any(0, keepdims=True), mg.shape)
            live = self.live_rows(np.concatenate([mp, mg.astype(np.float32)], 0), steps)
        video = self._dev(video, torch.float32)
        cap = self._dev(sampled, torch.int32)
        mask = self._dev(mask, torch.float32)
        gcap = self._dev(gt_caption, torch.int32)
        gmask = self._dev(gt_mask, torch.float32)
        B = video.shape[0]
        rep = cap.shape[0] // B
        assert gcap.shape[0] == B, "one ground-truth caption per video of the batch (reinforce_multitask_e2e_attribute_s2vt.py:977)"
        keep = self.dropout_rate if keep is None else keep
        lam = float(lambda_loss)
        N = (rep + 1) * B
        row_ids, video_attr = None, video
        if video_gt is not None:
            assert not reuse_sampler_state, "video_gt: the ground-truth rows read their own feature block, LSTM1 runs per row block"
            video_gt = self._dev(video_gt, torch.float32)
            assert video_gt.shape == video.shape
            row_ids = self._row_ids(B, rep + 1, video_base)
            video = torch.cat([video] * rep + [video_gt], 0).contiguous()      # one feature row per unrolled row
        fused = not dp.active() and self.n_caption_lstm_step <= 128
        if fused:
            # one process: both blocks' coefficients, the smoothing vector, the joined caption block and the two mask sums in ONE
            # library launch (the expressions of the branch below, in their order); the loss terms in one more
            coef, smooth_tm, cap_all, target_tm, sums = ops.mixed_prep(mask.contiguous(), gmask.contiguous(), self._dev(rewards, torch.float32), self._dev(baseline, torch.float32),
                                                            cap.contiguous(), gcap.contiguous(), lam, self.loss_weight, q1, smoothing, float(B * self.world_size))
            nll, _ = self._forward_loss(video, cap_all, coef, smooth_tm, 1 if row_ids else rep + 1, video_base, keep, reuse_sampler_state, steps=steps,
                                        live=live, target_tm=target_tm, row_ids=row_ids)
            losses = ops.mixed_loss(self._coef_used, nll, live, N, rep * B)
            loss_total = losses[2]
        else:
            adv = self._dev(rewards, torch.float32) - self._dev(baseline, torch.float32)
            sums = torch.stack([mask.sum(), gmask.sum()])
            dp.allreduce_small(sums)                                           # global sum(mask) of both objectives
            # (lambda enters as fp32 device scalars -- fl32(1 - lambda) rounded once from the double, fl32(lambda) -- and every
            #  division below is an fp32 tensor division: the very operations of s2vt_mixed_prep, so the fused single-process path
            #  and this data-parallel one give the same bits for any lambda, whatever the tensor library does with Python scalars)
            one_minus = torch.full((), 1.0 - lam, dtype=torch.float32, device=self.device)
            lam_t = torch.full((), lam, dtype=torch.float32, device=self.device)
            coef_pg = mask * (adv * one_minus)[:, None] / sums[0]              # [rep*B, Tc] policy gradient on the sampled captions
            if q1:                                                             # cross entropy on the ground truth (tf_s2vt.py:150-166, as xe_update)
                colsum = gmask.sum(0)
                dp.allreduce_small(colsum)
                coef_xe = (colsum[None, :] / float(B * self.world_size)).expand(B, -1) * self.loss_weight
            else:
                coef_xe = gmask * self.loss_weight
            coef_xe = coef_xe * (lam_t / sums[1])                              # [B, Tc]
            coef = torch.cat([coef_pg, coef_xe], 0).t().contiguous().view(-1)  # time-major over the (rep+1)*B rows
            smooth = torch.zeros((rep + 1) * B, dtype=torch.float32, device=self.device)
            smooth[rep * B:] = float(smoothing)
            smooth_tm = smooth.repeat(self.n_caption_lstm_step).contiguous()
            nll, _ = self._forward_loss(video, torch.cat([cap, gcap], 0).contiguous(), coef, smooth_tm, 1 if row_ids else rep + 1, video_base, keep,
                                        reuse_sampler_state, steps=steps, live=live, row_ids=row_ids)
            terms = self._coef_used * nll
            if live is None:
                per_row = terms.view(-1, N)
                loss_pg, loss_xe = per_row[:, :rep * B].sum(), per_row[:, rep * B:].sum()
            else:                                                    # (row of the unroll = live index % N: sampled rows first)
                is_pg = (live.long() % N) < rep * B
                loss_pg, loss_xe = terms[is_pg].sum(), terms[~is_pg].sum()
            loss_total = loss_pg + loss_xe
        self.backward(accumulate=False, overlap=False)
        attr_scale = attr_loss = None
        if true_labels is not None and self.label_dim > 0:
            attr_scale, attr_loss = self._attr_terms(video_attr, true_labels)
        one = getattr(self, "_one_over_world", None)                      # the bucket is already normalised: global "sum(mask)" = 1
        if one is None or one[1] != self.world_size:
            one = self._one_over_world = (torch.full((), 1.0 / self.world_size, device=self.device), self.world_size)
        self.apply_gradients(one[0], lr, clip_norm, weight_decay=lam * self.decay_value, attr_scale=attr_scale, decay_all=decay_all,
                             extra_sumsq=extra_sumsq)
        st = StepStats(loss_total, self._sumsq, sums[0])
        st.attr_loss = attr_loss
        return st

    def attribute_scores(self, video):
        """sigmoid(mean_t(video) . attr_W + attr_b) [B, label_dim] on the device -- the scores of evaluate_multilabel
        (reinforce_multitask_e2e_attribute_loss.py:621-624) for a feature block already in HBM."""
        if not self.label_dim:
            raise ValueError("attribute_scores / evaluate_multilabel need a model built with label_dim > 0")
        return ops.attr_head_scores(self._dev(video, torch.float32), self.store.p["attr_W"], self.store.p["attr_b"])[1]

    def evaluate_multilabel(self, threshold=0.5):
        """evaluate_multilabel(threshold) -> (video, scores) of reinforce_multitask_e2e_attribute_loss.py:606-626:
        scores = sigmoid(xw_plus_b(reduce_mean(video, axis=1), attr_W, attr_b)), [batch, label_num].  The reference's
        placeholder takes frames and runs the CNN in inference mode first (:608-620) -- so does this one once a CNN is attached
        (attach_cnn); with precomputed features the feed is the feature block [n, Tv, dim_image].  `threshold` is accepted and
        unused, exactly as there (the caller thresholds the scores)."""
        video = self._video_placeholder(None)

        def fn(v):
            return {"scores": self.attribute_scores(self._features(v)).cpu().numpy()}
        return video, Output("scores", fn, [video])



    def build_model(self, multilabel_normalised=False, with_multilabel_loss=False):
        """The cross-entropy graph.  Base class (label_dim == 0): (loss, video, caption, caption_mask, probs) as tf_s2vt.py:90-167.
        With an attribute head (label_dim > 0) the multitask classes' tuples:
          (loss, video_frames, caption, caption_mask, probs, true_labels)            reinforce_multitask_e2e_attribute_loss.py:116-226
          (..., probs, true_labels, multilabel_loss)  with_multilabel_loss=True       multitask_e2e_attribute_s2vt.py:116-223
        where loss = (1 - alpha) * XE / sum(mask) + weight_decay_loss + alpha * multilabel_loss, weight decay over EVERY
        trainable variable (Q3), and multilabel_loss = sum(bce) (:221 of the first script; multilabel_normalised=False) or
        sum(bce) / (label_dim * batch_size) (:219 of the second; True).  `video` is the frame placeholder once a CNN is attached
        (attach_cnn: the graph then starts at the frames, slim.dropout on the pooled features, :131).
        Fetching `loss` / `probs` evaluates the forward only; minimize() builds the train_op."""
        B, Tc = self.batch_size, self.n_caption_lstm_step
        video = self._video_placeholder(B)
        caption = Placeholder("caption", (B, Tc), np.int32)
        caption_mask = Placeholder("caption_mask", (B, Tc), np.float32)
        multitask = self.label_dim > 0
        true_labels = Placeholder("true_labels", (B, self.label_dim), np.float32) if multitask else None

        def fn(v, c, m, y=None):
            v = self._features(v, dropout=True, draw=0)
            c = self._dev(c, torch.int32); m = self._dev(m, torch.float32)
            n = c.shape[0]
            coef = ((m.sum(0)[:, None] / float(n)).expand(-1, n) * self.loss_weight).contiguous().view(-1)
            vid, sid = self._row_ids(v.shape[0], 1, 0)
            seed = self.dropout_seed + 104729 * self.global_step
            logits, _ = ops.teacher_forced_fwd(self.dims, self.store.params, v, c, n, self.dropout_rate, seed, vid, sid)
            probs = logits.view(Tc, n, -1).clone()
            nll, _ = ops.softmax_nll_fwd_bwd(logits, c.t().contiguous().view(-1), coef, 0.05)
            xe = float(torch.dot(coef, nll) / m.sum())
            out = {"probs": probs.cpu().numpy()}
            if multitask:
                _, _, bce = ops.attr_head_fwd(v, self.store.p["attr_W"], self.store.p["attr_b"], self._dev(y, torch.float32))
                ml = float(bce.sum()) / (float(self.label_dim * n) if multilabel_normalised else 1.0)
                out["multilabel_loss"] = ml
                out["loss"] = (1.0 - self.alpha) * xe + self.l2_term() + self.alpha * ml
            else:
                out["loss"] = xe + self.l2_term()
            return out
        inputs = [video, caption, caption_mask] + ([true_labels] if multitask else [])
        loss, probs = Output("loss", fn, inputs), Output("probs", fn, inputs)
        loss.graph = {"kind": "build_model", "inputs": inputs, "normalised": multilabel_normalised, "multilabel_loss": None}
        if not multitask:
            return loss, video, caption, caption_mask, probs
        ml = loss.graph["multilabel_loss"] = Output("multilabel_loss", fn, inputs)
        if with_multilabel_loss:
            return loss, video, caption, caption_mask, probs, true_labels, ml
        return loss, video, caption, caption_mask, probs, true_labels

    # ---- the nodes the reference's train() adds around the model's graphs
    def placeholder(self, name, shape=(None,), dtype=np.float32):
        """tf.placeholder: `rewards` / `base_line` of reinforcement_multisampling_tf_s2vt.py:628-629."""
        return Placeholder(name, shape, dtype)

    def exponential_decay(self, start_learning_rate, decay_steps, decay_rate=0.5):
        """tf.train.exponential_decay(lr, global_step, decay_steps, rate, staircase=True) on the model's own step counter
        (tf_s2vt.py:440-441, reinforcement_multisampling_tf_s2vt.py:638-640): a fetch (sess.run(learning_rate))
        with a .value() the train ops read."""
        def value():
            return float(start_learning_rate) * float(decay_rate) ** (self.global_step // int(decay_steps))
        out = Output("learning_rate", lambda: {"learning_rate": value()}, [])
        out.value = value
        return out

    def minimize(self, build_model_outputs, learning_rate, clip_norm=10.0):
        """train_op of tf_s2vt.py:442-445: AdamOptimizer(learning_rate).compute_gradients(tf_loss) ->
        clip_by_global_norm(10) -> apply_gradients(global_step) -- and of multitask_e2e_attribute_s2vt.py:715-717 /
        e2e_tf_s2vt.py:533-536, the same statement over the multitask loss and over CNN + captioner.  `build_model_outputs` =
        the tuple build_model() returned (5, 6 or 7 long); sess.run([train_op, tf_loss(, tf_multilabel_loss)], feed_dict) is ONE
        update, and the losses fetched beside it are the ones the update differentiated (same dropout masks, pre-update weights)."""
        loss = build_model_outputs[0]
        graph = getattr(loss, "graph", None) or {"inputs": list(build_model_outputs[1:4]), "normalised": True, "multilabel_loss": None}
        lr = learning_rate.value if hasattr(learning_rate, "value") else (lambda: float(learning_rate))

        def fn(v, c, m, y=None):
            wd = self.l2_term()                                                # the l2 term of tf_s2vt.py:163-166, pre-update
            if self._is_frames(v):
                st = self.e2e.xe_step(self._frames(v), c, m, lr(), clip_norm=clip_norm, true_labels=y, attr_normalised=graph["normalised"])
            else:
                st = self.xe_update(v, c, m, lr(), clip_norm=clip_norm, q1=True, smoothing=0.05, decay_all=self._decay_everything(),
                                    true_labels=y, attr_normalised=graph["normalised"])
            out = {"train_op": None, "loss": float(st.loss) + wd}
            if getattr(st, "attr_loss", None) is not None:
                out["loss"] += float(st.attr_loss)
                out["multilabel_loss"] = float(st.attr_loss.unscaled())
            return out
        provides = {loss: "loss"}
        if graph["multilabel_loss"] is not None:
            provides[graph["multilabel_loss"]] = "multilabel_loss"
        return Output("train_op", fn, graph["inputs"], provides=provides)

    def reinforce_train_op(self, build_loss_outputs, rewards, base_line, learning_rate, clip_norm=5.0):
        """(train_op, sum_loss) of reinforcement_multisampling_tf_s2vt.py:641-652:
            norm = sum(loss_masks); sum_loss = -sum(loss * (rewards - base_line)) / norm;
            clip_by_global_norm(tf.gradients(sum_loss), 5); Adam.apply_gradients(global_step).
        Fed as there (:821-823): {loss_masks, loss_captions, loss_features (the K-times tiled block), rewards, base_line}.
        (= multitask_train_op without an attribute term or a cross-entropy mix.)"""
        return self.multitask_train_op(build_loss_outputs[:4], rewards, base_line, learning_rate, clip_norm=clip_norm)

    def build_loss(self):
        """The REINFORCE graph: (loss, video, caption, caption_mask) as reinforcement_multisampling_tf_s2vt.py:227-292, and with an
        attribute head (label_dim > 0) the multitask class's (loss, video_frames, caption, caption_mask, true_labels,
        multilabel_loss) of reinforce_multitask_e2e_attribute_loss.py:228-380, multilabel_loss = sum(bce) / (label_dim * batch).
        The reference returns the dense [N,Tc,V] tensor log_softmax*onehot*mask; it has one non-zero per (n,t), so the fetch here is
        the [N,Tc] array of those values (lp * mask).  With a CNN attached `video` is the frame placeholder (slim.dropout on the
        pooled features with this graph's own mask, :250)."""
        N, Tc = self.batch_size * self.multisample, self.n_caption_lstm_step
        video = self._video_placeholder(N)
        caption = Placeholder("caption", (N, Tc), np.int32)
        caption_mask = Placeholder("caption_mask", (N, Tc), np.float32)

        def fn(v, c, m):
            v, B = self._loss_features(v, np.shape(c)[0])
            c = self._dev(c, torch.int32); m = self._dev(m, torch.float32)
            vid, sid = self._row_ids(B, c.shape[0] // B, 0)
            seed = self.dropout_seed + 104729 * self.global_step
            logits, _ = ops.teacher_forced_fwd(self.dims, self.store.params, v, c, c.shape[0], self.dropout_rate, seed, vid, sid)
            zero = torch.zeros(logits.shape[0], dtype=torch.float32, device=self.device)
            _, lp = ops.softmax_nll_fwd_bwd(logits, c.t().contiguous().view(-1), zero, 0.0)
            return {"loss": (lp.view(Tc, -1).t() * m).cpu().numpy()}
        loss = Output("loss", fn, [video, caption, caption_mask])
        loss.graph = {"kind": "build_loss"}
        if not self.label_dim:
            return loss, video, caption, caption_mask
        true_labels = Placeholder("true_labels", (N, self.label_dim), np.float32)

        def fn_ml(v, y):
            y = self._dev(y, torch.float32)
            v, _ = self._loss_features(v, y.shape[0])
            _, _, bce = ops.attr_head_fwd(v, self.store.p["attr_W"], self.store.p["attr_b"], y)
            return {"multilabel_loss": float(bce.sum()) / float(self.label_dim * v.shape[0])}
        return loss, video, caption, caption_mask, true_labels, Output("multilabel_loss", fn_ml, [video, true_labels])

    def _loss_features(self, v, N):
        """build_loss's video feed -> (the B distinct feature rows on the device, B): the K-times tiled block of
        reinforcement_multisampling_tf_s2vt.py:779-782 is recognised on the host (_untile); frames go through the CNN with the
        loss graph's own slim.dropout mask (draw 1; the sampler's is draw 0)."""
        if self._is_frames(v):
            f = self._frames(v)
            f, B = self._untile(f, N, host=None if f.is_cuda else f.numpy())
            return self._features(f, dropout=True, draw=1), B
        vh = None if isinstance(v, torch.Tensor) else v
        return self._untile(self._dev(v, torch.float32), N, host=vh)

    def multitask_train_op(self, build_loss_outputs, rewards, base_line, learning_rate, clip_norm=10.0, alpha=None,
                           build_model_outputs=None, lambda_loss=None):
        """(train_op, sum_loss) of the multitask / end-to-end REINFORCE scripts' train():

        * reinforce_multitask_e2e_attribute_loss.py:953-960 (build_loss_outputs = the 6-tuple, `alpha` given or the model's):
              norm = sum(loss_masks); residual = rewards - base_line
              sum_loss = -(1 - alpha) * sum(loss * residual) / norm + alpha * multilabel_loss
              clip_by_global_norm(tf.gradients(sum_loss, trainable_variables), 10); Adam.apply_gradients(global_step)
          fed {loss_masks, loss_captions, loss_features, rewards, base_line, true_labels} (:1113) -- the script also feeds
          model_features / model_captions / model_caption_masks, which sum_loss does not read; Session.run ignores them too.
        * reinforce_multitask_e2e_attribute_s2vt.py:846-855 (lambda_loss and build_model_outputs given; clip_norm=5 there):
              sum_loss = -(1 - lambda_loss) * sum(loss * residual) / norm + lambda_loss * model_loss
          with model_loss the cross-entropy graph on the ground-truth captions of the same videos (label smoothing, Q1, weight
          decay on every variable), fed additionally {model_features, model_captions, model_caption_masks} (:977).
        * a 4-tuple build_loss and no lambda: reinforcement_multisampling_tf_s2vt.py:641-652 (reinforce_train_op).

        Lowered onto ONE fused update each: reinforce_update(true_labels=) / mixed_update, or -- when the video feed is a frame
        block and a CNN is attached -- e2e.EndToEnd.reinforce_update / mixed_update (gradient through the CNN, one clip norm and one
        Adam over both halves).  sum_loss fetched beside train_op is the pre-update value the update differentiated."""
        video, caption, caption_mask = build_loss_outputs[1:4]
        true_labels = build_loss_outputs[4] if len(build_loss_outputs) >= 6 else None
        lr = learning_rate.value if hasattr(learning_rate, "value") else (lambda: float(learning_rate))
        mixed = lambda_loss is not None
        inputs = [video, caption, caption_mask, rewards, base_line]
        if true_labels is not None:
            inputs.append(true_labels)
        if mixed:
            if build_model_outputs is None:
                raise ValueError("lambda_loss mixes in model_loss: pass build_model_outputs (the tuple build_model() returned)")
            inputs += list(build_model_outputs[1:4])

        def fn(v, c, m, r, b, *rest):
            rest = list(rest)
            y = rest.pop(0) if true_labels is not None else None
            r = np.asarray(r, np.float32).reshape(-1); b = np.asarray(b, np.float32).reshape(-1)
            frames, v_fed = self._is_frames(v), v
            n_rows = np.shape(c)[0]
            if frames:
                f = self._frames(v)
                v, _ = self._untile(f, n_rows, host=None if f.is_cuda else f.numpy())
            else:
                vh = None if isinstance(v, torch.Tensor) else v
                v, _ = self._untile(self._dev(v, torch.float32), n_rows, host=vh)
            if y is not None and np.shape(y)[0] != v.shape[0]:
                y = np.asarray(y, np.float32)[:v.shape[0]]                    # labels arrive tiled like the features
            saved = self.alpha
            if alpha is not None:
                self.alpha = alpha
            try:
                if not mixed:
                    upd = self.e2e.reinforce_update if frames else self.reinforce_update
                    st = upd(v, c, m, r, b, lr(), clip_norm=clip_norm, true_labels=y)
                    total = float(st.loss)
                else:
                    mv, mc, mm = rest
                    wd = float(lambda_loss) * self.l2_term()                   # lambda * weight_decay_loss inside model_loss, pre-update
                    same = mv is v_fed                                       # the script feeds the one video_batch to both graphs (:977)
                    if not same and not isinstance(mv, torch.Tensor) and not isinstance(v_fed, torch.Tensor):
                        mva = np.asarray(mv, np.float32)
                        same = np.array_equal(mva, np.asarray(v_fed, np.float32)[:mva.shape[0]])
                    if frames:
                        if not same:
                            raise ValueError("model_features and loss_features must be the same frame batch (the script feeds video_batch to both, :977)")
                        st = self.e2e.mixed_update(v, c, m, r, b, mc, mm, lr(), lambda_loss=lambda_loss, clip_norm=clip_norm, true_labels=y)
                    else:
                        st = self.mixed_update(v, c, m, r, b, mc, mm, lr(), lambda_loss=lambda_loss, clip_norm=clip_norm, true_labels=y,
                                               decay_all=self._decay_everything(), video_gt=None if same else self._dev(mv, torch.float32))
                    total = float(st.loss) + wd
            finally:
                self.alpha = saved
            if getattr(st, "attr_loss", None) is not None:
                total += float(st.attr_loss)
            return {"train_op": None, "sum_loss": total}
        return Output("train_op", fn, inputs), Output("sum_loss", fn, inputs)
