"""Pieces shared by the training drivers (train_xe.py / train_rl.py / train_e2e.py): configuration, corpus loading,
the epoch iterator of the reference's train() loops, greedy evaluation, checkpoints, and the data-parallel frame:
one process per GPU under torch.distributed.run, every rank walks the SAME shuffled epoch (same seed) and takes its
contiguous shard of each global batch (dist.shard_range); noise counters carry global video indices, so n ranks x B/n
train exactly as one rank x B does."""
from __future__ import annotations

import os
import random
from dataclasses import dataclass

import numpy as np

from . import data, hostglue, reward
from . import dist as dp


@dataclass
class Config:
    # the "Train Parameters" blocks of tf_s2vt.py:304-321 / reinforcement_multisampling_tf_s2vt.py:505-517
    dim_image: int = 1536
    lstm_dim: int = 1000
    word_dim: int = 500
    n_video_lstm_step: int = 5
    n_caption_lstm_step: int = 35
    n_epochs: int = 30
    batch_size: int = 64
    start_learning_rate: float = 1e-3
    decay_steps: int = 5000            # tf.train.exponential_decay(lr, step, decay_steps, 0.5, staircase=True)
    clip_norm: float = 10.0
    multisample: int = 8               # RL: sampler passes per step (K)
    seed: int = 4
    model_path: str = "./models"
    model_name: str = "s2vt_model"
    max_steps_per_epoch: int = 0       # 0 = the whole epoch (tests bound it)
    checkpoint_format: str = "npz"     # "npz" (name -> array dump), "tf" (a TensorFlow checkpoint in the format the mirrored script's own Saver writes:
                                       # V2 -- <name>-<epoch>.index / .data-00000-of-00001 -- for the REINFORCE / e2e / attention drivers, ONE V1 file
                                       # <name>-<epoch> for train_xe, whose reference saver is write_version=1, tf_s2vt.py:440) or "tf_v1" (V1 everywhere).
                                       # TF-format files are a SUPERSET of the reference's: besides its names they carry the Adam slots / beta powers /
                                       # counter (the XE reference saver holds the model variables only) and one private int64 '_s2vt/adam_t'.
    step_log: str = ""                 # path of a JSONL step log ("" = none)
    stop_at_eos: bool = False          # RL: samples leave the decode loop at their first <eos> (opt-in; the reference samples all Tc steps and masks
                                       # afterwards -- same update, shorter loop: model.sample(stop_at_eos=True))
    alpha: float = 0.05                # multitask scripts: weight of the attribute head's multilabel loss (reinforce_multitask_e2e_attribute_loss.py:697)
    lambda_loss: float = 0.0           # multitask scripts: weight of the ground-truth XE term mixed into the REINFORCE objective
                                       # (reinforce_multitask_e2e_attribute_s2vt.py:670,850: 0.5 there); 0 = the pure self-critical objective


class Corpus:
    def __init__(self, sent_file, feature_file, vocabulary=None, vocabulary_file=None):
        self.captions, self.features = data.get_video_feature_caption_pair(sent_file, feature_file)
        if vocabulary is None and vocabulary_file is not None:
            vocabulary = data.read_vocabulary(vocabulary_file)
        self.vocabulary = vocabulary
        self.index = data.CaptionIndex(self.captions)


class StepLog:
    """One JSON object per line per training step / epoch (the reference prints and appends to loss.txt, tf_s2vt.py:463-499):
    machine-readable, flushed per record, safe to tail.  path=None disables it."""

    def __init__(self, path=None):
        self._f = open(path, "a") if path else None

    def write(self, **record):
        if self._f:
            import json
            self._f.write(json.dumps(record) + "\n")
            self._f.flush()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None


def learning_rate(cfg: Config, global_step: int) -> float:
    return cfg.start_learning_rate * 0.5 ** (global_step // cfg.decay_steps)


def epoch_batches(n_items: int, batch_size: int, rng: random.Random):
    """The reference's iteration: shuffle all (video, sentence) pairs, walk full batches only
    (zip(range(0, n - B, B), range(B, n, B)), tf_s2vt.py:477-482)."""
    index = list(range(n_items))
    rng.shuffle(index)
    for start, end in zip(range(0, n_items - batch_size, batch_size), range(batch_size, n_items, batch_size)):
        yield index[start:end]


class DataParallel:
    """The data-parallel frame of a training driver.  world == 1 (no WORLD_SIZE in the environment) is the plain
    single-GPU run of the reference."""

    def __init__(self, device=None):
        import torch
        if device is None and not torch.cuda.is_available():
            raise RuntimeError("the training drivers need a GPU: the HIP library has no CPU fallback")
        self.rank, self.world, self.device = dp.init_from_env(device)

    @property
    def chief(self) -> bool:
        return self.rank == 0

    def per_rank(self, global_batch: int) -> int:
        assert global_batch % self.world == 0, f"batch_size {global_batch} must divide over {self.world} ranks"
        return global_batch // self.world

    def shard(self, idx):
        """This rank's contiguous slice of one global batch and the GLOBAL index of its first video."""
        lo, hi = dp.shard_range(len(idx), self.rank, self.world)
        return idx[lo:hi], lo

    def attach(self, model):
        model.world_size, model.rank = self.world, self.rank
        return model

    def mean(self, total: float, count: float):
        """Mean over all ranks of per-rank (sum, count) pairs."""
        import torch
        if self.world == 1:
            return total / count if count else None
        t = torch.tensor([total, count], dtype=torch.float64, device=self.device)
        dp.allreduce_small(t)
        return float(t[0] / t[1]) if float(t[1]) else None

    def sum_ints(self, counts):
        """Element-wise sum over all ranks of a small integer vector (the multilabel evaluation's confusion counts)."""
        import torch
        if self.world == 1:
            return np.asarray(counts, np.int64)
        t = torch.as_tensor(np.asarray(counts, np.float64), device=self.device)
        dp.allreduce_small(t)
        return t.cpu().numpy().round().astype(np.int64)

    def gather_dict(self, d: dict) -> dict:
        if self.world == 1:
            return d
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, d)
        out = {}
        for q in parts:
            out.update(q)
        return out


def lookahead(iterable):
    """(item, has_next, peek) triples of an iterable: the training loops prepare batch i + 1 while the GPU works on batch i."""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def run_step(model, fn, log=print, retries: int = 1, overlap=None):
    """One training step with the persistent-recurrence guard: fn() runs the step and returns its statistics; the host then
    reads the loss (the synchronisation point the reference's sess.run is) and checks the library's health.  On
    S2VTChainTimeout -- a persistent LSTM recurrence was starved of CUs, e.g. by another process on the GPU -- the
    variables of THIS process are intact (Adam launches behind the fault skip on the device).
    Single process: recover() rewinds the step counter to the last applied update, switches to per-step launches, and the
    SAME batch is repeated.
    Data parallel (world > 1): the fault is local to one rank but the step is collective -- the faulting rank's garbage
    gradients have already taken part in the all-reduce (the healthy ranks applied them) or, if the fault surfaced earlier,
    its peers are waiting in a collective it will never join; repeating the batch on one rank would pair its all-reduces
    with the peers' NEXT batch.  So no retry: the exception propagates, the process exits non-zero, torch.distributed.run
    tears the job down, and the launcher restarts it from the last checkpoint (`--resume`; checkpoints are only written
    after a collective health check, save_checkpoint_checked).
    overlap: host work that does not depend on this step's result (preparing the next batch, writing the previous step's
    log lines); it runs once, after the step has been queued on the GPU and before the host blocks on its loss."""
    from ._lib import S2VTChainTimeout
    collective = getattr(model, "world_size", 1) > 1
    for attempt in range(retries + 1):
        try:
            st = fn()
            if overlap is not None:
                overlap, run = None, overlap
                run()
            loss = float(st.loss)                      # device -> host: everything queued for this step has run
            model.check_health()
            return st, loss
        except S2VTChainTimeout as e:
            if collective:
                log(f"persistent recurrence timed out on rank {getattr(model, 'rank', '?')} of a data-parallel job ({e}); replicas can no "
                    "longer be kept in step from here: exiting so that the launcher restarts from the last checkpoint")
                raise
            if attempt == retries:
                raise
            step, lost = model.recover()
            log(f"persistent recurrence timed out ({e}); variables intact at step {step} ({lost} update(s) skipped); "
                "continuing with per-step launches and repeating the batch")
    raise AssertionError("unreachable")


def all_ranks_healthy(model) -> bool:
    """Collective (every rank must call it): True when NO rank has a persistent-recurrence fault pending.  One 1-element MAX
    all-reduce; a plain local check in a single process."""
    from . import ops
    bad = 1.0 if ops.chain_fault() else 0.0
    if getattr(model, "world_size", 1) > 1 and dp.active():
        import torch
        import torch.distributed as dist
        t = torch.tensor([bad], dtype=torch.float32, device=model.device)
        if t.is_cuda and dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = float(t[0])
    return bad == 0.0


def greedy_eval(model, corpus: Corpus, ixtoword, scorer: "reward.CiderD | None", batch_size: int, par: "DataParallel | None" = None):
    """Greedy captions for every test video (tf_s2vt.py:508-524) and their mean CIDEr-D against the video's own
    references when a scorer over that corpus is given (the reference reports BLEU/METEOR/ROUGE/CIDEr through
    the external coco-caption package, which is not part of this build).  Data parallel: the videos are dealt
    round-robin over the ranks, captions gathered, the score averaged over all of them."""
    vids = corpus.index.video_ids
    if par is not None and par.world > 1:
        vids = vids[par.rank::par.world]
    decoded, total, count = {}, 0.0, 0
    for a in range(0, len(vids), batch_size):
        ids = vids[a:a + batch_size]
        _, g = model.sample(corpus.features.batch(ids), 0, True)
        g = g.cpu().numpy()
        model.check_health()
        for v, s in zip(ids, hostglue.decode_captions(g, ixtoword)):
            decoded[v] = s
        if scorer is not None:
            sc = scorer.score_ids(g, [corpus.index.row[v] for v in ids])
            total += float(sc.sum()); count += len(sc)
    if par is not None and par.world > 1:
        return par.gather_dict(decoded), (par.mean(total, count) if scorer is not None else None)
    return decoded, (total / count if count else None)


def save_checkpoint(model, cfg: Config, epoch: int, step_name: str = "g_step", tf_version: int = 2):
    """Variables under the reference's TF names plus what its tf.train.Saver keeps beside them when it is created after
    the optimizer (reinforcement_multisampling_tf_s2vt.py:661): Adam slots, beta powers, the step counter -- a resumed
    run continues the moments, the bias correction, the learning-rate staircase and the noise streams."""
    os.makedirs(cfg.model_path, exist_ok=True)
    tf_fmt = cfg.checkpoint_format in ("tf", "tf_v1")
    sd = model.store.state_dict(global_step=model.global_step, adam_t=model.adam_t, step_name=step_name,
                                counter_dtype=np.int32 if tf_fmt else np.int64)      # the graph's counter is tf.Variable(0): DT_INT32
    if tf_fmt:                                             # the files tf.train.Saver.save(sess, path, global_step=epoch) leaves (:661)
        from . import tfckpt
        path = os.path.join(cfg.model_path, f"{cfg.model_name}-{epoch}")
        sd.pop("global_step", None)                        # (this repository's alias; the graph's counter is `step_name`)
        if "adam_t" in sd:                                 # not a variable of the reference's graph: kept under a clearly private name
            sd["_s2vt/adam_t"] = sd.pop("adam_t")
        arrays = {k: np.asarray(v) for k, v in sd.items()}
        if cfg.checkpoint_format == "tf_v1" or tf_version == 1:    # tf_s2vt.py:440: tf.train.Saver(write_version=1) -- ONE file, the V1 tensor-slice format
            tfckpt.write_checkpoint_v1(path, arrays)
        else:
            tfckpt.write_checkpoint_v2(path, arrays)
        return path
    path = os.path.join(cfg.model_path, f"{cfg.model_name}-{epoch}.npz")
    np.savez(path, **sd)
    return path


def save_checkpoint_checked(model, cfg: Config, epoch: int, step_name: str = "g_step", chief: bool = True, tf_version: int = 2):
    """save_checkpoint behind a COLLECTIVE health check (every rank calls this; the chief writes): a rank whose
    persistent recurrence faulted has fed garbage into an all-reduce, so no replica's variables may be written out."""
    if not all_ranks_healthy(model):
        from ._lib import S2VTChainTimeout
        raise S2VTChainTimeout("a rank of this data-parallel job has a persistent-recurrence fault pending: not writing a checkpoint")
    return save_checkpoint(model, cfg, epoch, step_name, tf_version) if chief else None


def optimistic_restore(model, path, restore_step: bool = True, step_names=("global_step", "g_step", "Variable"), optimizer_state="auto"):
    """Load every variable whose name and shape match (reinforcement_multisampling_tf_s2vt.py:47-61) from an .npz dump or a
    TensorFlow checkpoint FILE (V2 `<prefix>.index` + data shards, or a V1 file: tfckpt.py) -- Adam slots and
    beta powers included, as there -- and position the model's counters: the step counter when the checkpoint holds one
    under a name this run's graph would have (`step_names`: a REINFORCE run started from an XE checkpoint passes
    ('g_step',) and so starts its staircase at 0, as the reference does), Adam's update count from beta1_power."""
    from . import tfckpt
    raw = tfckpt.read_checkpoint(path)                     # .npz dump, TensorFlow V2 prefix (<path>.index) or V1 file
    sd = {k: v for k, v in raw.items() if k in step_names or k not in ("global_step", "g_step", "Variable")}
    # optimizer_state "auto": Adam's slots / beta powers / update count are taken only from a checkpoint of a run whose graph
    # had them under THIS run's counter -- i.e. one that carries a counter named in `step_names`.  The reference's XE saver
    # (tf_s2vt.py:440) is created BEFORE the optimizer and holds the model variables only, so a REINFORCE run restored from an XE
    # checkpoint starts Adam from zero moments (optimistic_restore finds no slots, :47-61); our XE checkpoints do carry the
    # slots (for --resume), and loading them here would drive the first REINFORCE updates with XE-scale moments.
    keep_opt = optimizer_state is True or (optimizer_state == "auto" and any(k in raw for k in step_names))
    if not keep_opt:
        sd = {k: v for k, v in sd.items() if not (k.endswith("/Adam") or k.endswith("/Adam_1") or k in ("beta1_power", "beta2_power", "adam_t", "_s2vt/adam_t"))}
    loaded = model.store.load_state_dict(sd)
    st = model.store
    if restore_step and (st.restored_step is not None or st.restored_adam_t is not None):
        step = st.restored_step if st.restored_step is not None else 0
        model.set_step(step, st.restored_adam_t if st.restored_adam_t is not None else step)
    return loaded
