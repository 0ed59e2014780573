"""Host-side logic that needs no GPU: the sess.run shim's evaluation order, checkpoints under the TF variable names with the
Adam slots tf.train.Saver keeps, the JSONL step log, the data-parallel switches."""
import json
import os

import numpy as np
import torch


def test_session_run_evaluates_updates_first_and_shares_their_loss():
    import s2vt_amd
    from s2vt_amd.model import Output, Placeholder, Session
    x = Placeholder("x", (None,), np.float32)
    state = {"w": 1.0, "evals": []}

    def forward(v):
        state["evals"].append("fwd")
        return {"loss": state["w"] * float(np.sum(v)), "probs": np.asarray(v) * state["w"]}

    loss = Output("loss", forward, [x])
    probs = Output("probs", forward, [x])

    def update(v):                                   # a train_op: computes the loss it differentiates, then changes the variable
        state["evals"].append("upd")
        l = state["w"] * float(np.sum(v))
        state["w"] *= 0.5
        return {"train_op": None, "loss": l}

    train_op = Output("train_op", update, [x], provides={loss: "loss"})
    sess = Session(None)
    feed = {x: np.array([1.0, 2.0], np.float32)}
    # sess.run([train_op, tf_loss]): ONE evaluation, and the loss is the pre-update one (TF: train_op depends on the loss node)
    out = sess.run([train_op, loss], feed)
    assert out == [None, 3.0] and state["evals"] == ["upd"] and state["w"] == 0.5
    # order in the fetch list does not matter
    out = sess.run([loss, train_op], feed)
    assert out == [1.5, None] and state["evals"] == ["upd", "upd"]
    # two fetches of one graph call share it; a fetch the update does not provide is evaluated after it
    state["evals"].clear()
    l, p = sess.run([loss, probs], feed)
    assert state["evals"] == ["fwd"] and l == 0.25 * 3.0 and np.allclose(p, [0.25, 0.5])
    assert sess.run(loss, feed) == 0.75                                  # a single fetch returns the bare value
    try:
        sess.run(loss, {})
        assert False
    except KeyError as e:
        assert "x" in str(e)                                             # a missing feed names the placeholder


def test_checkpoint_keeps_tf_names_adam_slots_and_step():
    import s2vt_amd
    from s2vt_amd import model as M
    shapes = M.param_shapes(8, 11, 4, 4, label_dim=3)
    a = M.ParamStore(shapes, torch.device("cpu"))
    M.init_reference(a, seed=1)
    a.m.uniform_(-1, 1); a.v.uniform_(0, 1)
    sd = a.state_dict(global_step=7)
    assert "s2vt/LSTM1/basic_lstm_cell/weights" in sd and "s2vt/LSTM2/basic_lstm_cell/biases/Adam_1" in sd and "Wemb/Adam" in sd
    assert int(sd["global_step"]) == 7 and abs(float(sd["beta1_power"]) - 0.9 ** 8) < 1e-7
    b = M.ParamStore(shapes, torch.device("cpu"))
    loaded = b.load_state_dict(sd)
    assert b.restored_step == 7 and b.restored_adam_t == 7
    assert len(loaded) == 3 * len(a.names) + 4                          # + global_step, g_step (the REINFORCE script's name), beta1_power, adam_t
    # a REINFORCE run started from an XE checkpoint: Adam's slots and beta powers match by name, the unnamed counter
    # ('Variable', tf_s2vt.py:441) is a different variable from 'g_step' (reinforcement_multisampling_tf_s2vt.py:637)
    sdx = a.state_dict(global_step=12, adam_t=12, step_name="Variable")
    assert int(sdx["Variable"]) == 12 and "g_step" not in sdx
    d = M.ParamStore(shapes, torch.device("cpu"))
    d.load_state_dict({k: v for k, v in sdx.items() if k not in ("global_step", "Variable")})
    assert d.restored_step is None and d.restored_adam_t == 12
    e = M.ParamStore(shapes, torch.device("cpu"))
    e.load_state_dict({"Variable": np.int64(5)})                         # the reference's own name for the counter is understood
    assert e.restored_step == 5
    for n in a.names:
        assert torch.equal(a.p[n], b.p[n]) and torch.equal(a._view(a.m, n), b._view(b.m, n)) and torch.equal(a._view(a.v, n), b._view(b.v, n))
    # optimistic_restore: a variable of another shape is skipped, unknown names are ignored, variables-only dumps load too
    sd2 = {k: v for k, v in a.state_dict().items()}
    sd2["Wemb"] = np.zeros((5, 5), np.float32); sd2["something/else"] = np.zeros(3, np.float32)
    c = M.ParamStore(shapes, torch.device("cpu"))
    names = c.load_state_dict(sd2)
    assert "Wemb" not in names and "something/else" not in names and "encode_image_W" in names and c.restored_step is None


def test_session_run_evaluates_plain_fetches_before_state_changing_ones():
    """sess.run([train_op, loss, probs]): in TF all three come from the same pre-update pass.  Here `loss` is a by-product
    of the train op; `probs` (nobody provides it) must be evaluated BEFORE the update runs."""
    import s2vt_amd
    from s2vt_amd import model as M
    x = M.Placeholder("x", (None,), np.float32)
    state = {"w": 1.0, "order": []}

    def fwd(v):
        state["order"].append("fwd")
        return {"loss": state["w"] * 10.0, "probs": state["w"]}

    def upd(v):
        state["order"].append("update")
        l = state["w"] * 10.0
        state["w"] += 1.0
        return {"train_op": None, "loss": l}
    loss, probs = M.Output("loss", fwd, [x]), M.Output("probs", fwd, [x])
    train_op = M.Output("train_op", upd, [x], provides={loss: "loss"})
    _, l, p = M.Session(None).run([train_op, loss, probs], {x: 0})
    assert state["order"] == ["fwd", "update"] and l == 10.0 and p == 1.0       # probs saw w = 1 (pre-update), one forward for it
    state["order"].clear()
    _, l = M.Session(None).run([train_op, loss], {x: 0})
    assert state["order"] == ["update"] and l == 20.0                            # nothing to evaluate early: ONE pass


def test_step_log_is_one_json_object_per_line(tmp_path):
    import s2vt_amd
    from s2vt_amd.train_common import StepLog
    log = StepLog(str(tmp_path / "s.jsonl"))
    log.write(kind="step", step=1, loss=0.5)
    log.write(kind="epoch", epoch=0, loss=0.4, ciderD=None)
    log.close()
    recs = [json.loads(l) for l in open(tmp_path / "s.jsonl")]
    assert recs == [{"kind": "step", "step": 1, "loss": 0.5}, {"kind": "epoch", "epoch": 0, "loss": 0.4, "ciderD": None}]
    StepLog(None).write(kind="step")                                     # disabled: a no-op


def test_data_parallel_switches_without_a_process_group(monkeypatch):
    import s2vt_amd
    from s2vt_amd import dist as dp
    assert dp.world_size() == 1 and not dp.active()
    monkeypatch.setenv("S2VT_DP_FORCE", "1")
    assert not dp.active()                                               # forcing needs an initialised process group
    t = torch.arange(4.0)
    g = torch.zeros(8)
    out = dp.allreduce_bucket(g, 4, 3.0)                                 # no group: the bucket is untouched, sum(mask) rides in the tail
    assert float(out) == 3.0 and dp.allreduce_async(t) is None and torch.equal(dp.allreduce_small(t), torch.arange(4.0))
    assert dp.shard_range(64, 3, 8) == (24, 32)


def test_active_steps_counts_the_live_leading_steps():
    """Video_Caption_Generator.active_steps: the steps behind the longest caption of a batch are not unrolled."""
    from s2vt_amd import hostglue
    from s2vt_amd.model import Video_Caption_Generator as G
    ids = np.array([[3, 4, 0, 0, 3, 0], [3, 3, 3, 0, 9, 9], [0, 4, 4, 4, 4, 4]], np.int32)
    mask = hostglue.masks_from_ids(ids)                                   # 1 up to and including the first <eos>
    assert G.active_steps(mask) == 4
    assert G.active_steps(torch.as_tensor(mask)) == 4
    assert G.active_steps(np.zeros((2, 5), np.float32)) == 1
    assert G.active_steps(np.ones((2, 5), np.float32)) == 5
    assert G.active_steps(None) is None


def test_adam_count_is_stored_outright_and_survives_a_denormal_beta_power():
    """beta1_power = 0.9^(t+1) underflows fp32 past t ~ 800; the checkpoint states Adam's update count as an integer."""
    import s2vt_amd
    from s2vt_amd import model as M
    shapes = M.param_shapes(8, 11, 4, 4)
    a = M.ParamStore(shapes, torch.device("cpu"))
    sd = a.state_dict(global_step=5000, adam_t=4321)
    assert int(sd["adam_t"]) == 4321 and float(sd["beta1_power"]) == 0.0
    b = M.ParamStore(shapes, torch.device("cpu"))
    b.load_state_dict(sd)
    assert b.restored_step == 5000 and b.restored_adam_t == 4321
    # a TF-format dump stores the graph's counter as DT_INT32 (tf.Variable(0, trainable=False))
    sd32 = a.state_dict(global_step=9, step_name="Variable", counter_dtype=np.int32)
    assert sd32["Variable"].dtype == np.int32


def test_restore_into_reinforce_drops_the_xe_runs_adam_state(tmp_path):
    """`train_rl --restore <XE checkpoint>` -- the reference's XE saver holds the model variables only
    (tf_s2vt.py:440), so its REINFORCE run starts Adam from zero moments; our XE checkpoints carry the slots for --resume and
    the restore path must not load them.  A checkpoint of the REINFORCE driver itself ('g_step') keeps them."""
    import s2vt_amd
    from s2vt_amd import model as M, train_common as TC

    class Stub:
        def __init__(self):
            self.store = M.ParamStore(M.param_shapes(8, 11, 4, 4), torch.device("cpu"))
            self.global_step = self.adam_t = 0

        def set_step(self, g, t=None):
            self.global_step, self.adam_t = int(g), int(g if t is None else t)

    src = Stub()
    M.init_reference(src.store, seed=2)
    src.store.m.uniform_(-1, 1); src.store.v.uniform_(0, 1)
    xe = tmp_path / "xe.npz"
    np.savez(xe, **src.store.state_dict(global_step=37, adam_t=37, step_name="Variable"))
    dst = Stub()
    loaded = TC.optimistic_restore(dst, str(xe), step_names=("g_step",))
    assert all(torch.equal(src.store.p[n], dst.store.p[n]) for n in src.store.names)
    assert float(dst.store.m.abs().max()) == 0.0 and float(dst.store.v.abs().max()) == 0.0      # fresh Adam
    assert dst.global_step == 0 and dst.adam_t == 0 and not any(k.endswith("/Adam") for k in loaded)
    rl = tmp_path / "rl.npz"
    np.savez(rl, **src.store.state_dict(global_step=12, adam_t=12, step_name="g_step"))
    dst2 = Stub()
    TC.optimistic_restore(dst2, str(rl), step_names=("g_step",))
    same_m = lambda a, b: all(torch.equal(a.store._view(a.store.m, n), b.store._view(b.store.m, n)) for n in a.store.names)
    assert same_m(dst2, src) and dst2.global_step == 12 and dst2.adam_t == 12
    dst3 = Stub()                                                                                  # --resume of the XE driver keeps everything
    TC.optimistic_restore(dst3, str(xe))
    assert same_m(dst3, src) and dst3.global_step == 37


def test_run_step_retries_alone_but_never_inside_a_data_parallel_job():
    """a persistent-recurrence fault is local to one rank, the step is collective: under world > 1 the driver must not
    repeat the batch on its own (its all-reduces would pair with the peers' next batch) -- the exception propagates."""
    import pytest
    import s2vt_amd
    from s2vt_amd import train_common as TC
    from s2vt_amd._lib import S2VTChainTimeout

    class Fake:
        def __init__(self, world):
            self.world_size, self.rank, self.calls, self.recovered = world, 1, 0, 0

        def check_health(self):
            pass

        def recover(self):
            self.recovered += 1
            return 0, 1

    class St:
        loss = 1.5

    def step(m):
        def fn():
            m.calls += 1
            if m.calls == 1:
                raise S2VTChainTimeout("starved")
            return St()
        return fn
    solo = Fake(1)
    st, loss = TC.run_step(solo, step(solo), log=lambda *_: None)
    assert loss == 1.5 and solo.calls == 2 and solo.recovered == 1
    dp2 = Fake(2)
    with pytest.raises(S2VTChainTimeout):
        TC.run_step(dp2, step(dp2), log=lambda *_: None)
    assert dp2.calls == 1 and dp2.recovered == 0


def test_untile_checks_host_feeds_on_the_host():
    """a [N, ...] feed whose K blocks are NOT copies of one another is N distinct videos, not a tiling."""
    import s2vt_amd
    from s2vt_amd import model as M
    m = M.Video_Caption_Generator(6, 11, 4, 4, 2, 0, 3, 4, device="cpu", multisample=3)
    rng = np.random.default_rng(0)
    base = rng.standard_normal((2, 3, 6)).astype(np.float32)
    tiled = np.tile(base, (3, 1, 1))
    v, B = m._untile(torch.as_tensor(tiled), 6, host=tiled)
    assert B == 2 and torch.equal(v, torch.as_tensor(base))
    distinct = rng.standard_normal((6, 3, 6)).astype(np.float32)
    v, B = m._untile(torch.as_tensor(distinct), 6, host=distinct)
    assert B == 6 and v.shape[0] == 6
    v, B = m._untile(torch.as_tensor(distinct), 6)               # a device feed is taken at the feed contract's word
    assert B == 2


# ---- round 5 ---------------------------------------------------------------------------------------------------------------------
def test_single_process_bucket_exchange_copies_nothing():
    """dist.allreduce_bucket in ONE process: the local sum(mask) IS the global one -- it is returned as given (no 4-byte copy into the
    bucket's tail: that was the last non-library launch in the XE step's trace), the tail view when the caller says the slot holds it."""
    from s2vt_amd import dist as dp
    g = torch.arange(8, dtype=torch.float32)
    tail0 = float(g[6])
    ms = torch.tensor([5.0])
    out = dp.allreduce_bucket(g, 6, ms)
    assert out.shape == (1,) and out.data_ptr() == ms.data_ptr() and float(g[6]) == tail0
    out = dp.allreduce_bucket(g, 6, torch.tensor(7.0))                # 0-dim tensors (mixed_update's 1 / world) come back with one element
    assert out.shape == (1,) and float(out) == 7.0
    out = dp.allreduce_bucket(g, 6, None)
    assert out.data_ptr() == g[6:7].data_ptr()
    assert float(dp.allreduce_bucket(g, 6, 3.0)) == 3.0


def test_lazy_scalar_is_computed_once_and_only_when_read():
    from s2vt_amd.model import LazyScalar
    n = {"calls": 0}

    def fn():
        n["calls"] += 1
        return torch.tensor(2.5)
    x = LazyScalar(fn)
    assert n["calls"] == 0
    assert float(x) == 2.5 and x.item() == 2.5 and float(x.tensor()) == 2.5 and n["calls"] == 1


def test_oracle_decay_all_and_attribute_scores():
    """The multitask / e2e scripts' always-true weight-decay predicate (reinforce_multitask_e2e_attribute_s2vt.py:222): decay_all adds
    exactly decay * (l2_loss(lstm1_b) + l2_loss(lstm2_b)) in both oracles; evaluate_multilabel's scores are sigmoid(z) of the head's z."""
    from oracle import s2vt_oracle as orc
    from oracle import s2vt_torch as T
    d = orc.Dims(24, 61, 8, 12, 3, 4, 5)
    p = orc.init_params(d, seed=2, attr=True)
    rng = np.random.default_rng(0)
    for k in ("lstm1_b", "lstm2_b"):
        p[k] = rng.uniform(-.3, .3, p[k].shape).astype(np.float32)
    video = np.abs(rng.standard_normal((3, 3, 24))).astype(np.float32)
    cap = rng.integers(0, 61, (3, 4)).astype(np.int32)
    mask = np.ones((3, 4), np.float32)
    logits = orc.teacher_forced(p, d, video, cap)
    extra = 5e-5 * 0.5 * float((p["lstm1_b"].astype(np.float64) ** 2).sum() + (p["lstm2_b"].astype(np.float64) ** 2).sum())
    a, b = orc.xe_loss(p, d, logits, cap, mask), orc.xe_loss(p, d, logits, cap, mask, decay_all=True)
    assert extra > 1e-8 and abs((b - a) - extra) <= 1e-12 + 1e-9 * extra
    pt = T.to_torch(p, torch.float64, False)
    lt = torch.as_tensor(logits).double()
    ta, tb = T.xe_loss(pt, lt, cap, mask), T.xe_loss(pt, lt, cap, mask, decay_all=True)
    assert abs(float(tb - ta) - extra) <= 1e-12 + 1e-9 * extra
    z, _ = orc.attr_head(p, video)
    sc = orc.attr_scores(p, video)
    assert sc.shape == (3, 5) and np.abs(sc - 1.0 / (1.0 + np.exp(-z.astype(np.float64)))).max() <= 2e-7


def test_bench_workload_table_and_profile_keys():
    """bench.py's workload table (the reference's own default configuration is a workload) and the rocprofv3-name -> class:tile mapping
    the stamped traffic / SQ summaries are keyed by."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench
    from prof_keys import prof_key
    w = bench.WORKLOADS["rl_ref"]
    assert (w["B"], w["K"], w["tc"], w["v"]) == (256, 8, 35, 9972) and w["tokens"](256, 8) == 256 * 8 * 35 and w["seqfwd"](256, 8) == 33 * 256
    assert bench.WORKLOADS["rl"]["tokens"](64, 5) == 6400 and abs(bench.F_SEQ - 1187.68e6) < 1
    assert len(bench.kernel_signature()) == 16
    assert prof_key("void s2vt::gemm_tn_dma_kernel<false, 16, 2>(s2vt::TnKArgs)") == "3:tn128x128(dma)"
    assert prof_key("void s2vt::gemm_kernel<2, 2, 2, 3, 1, 2, true, 32, 0, false, false>(s2vt::GemmArgs)") == "2:64x96(2x2)"
    assert prof_key("void s2vt::gemm_kernel<1, 4, 6, 1, 4, 3, true, 32, 4, false, false>(s2vt::GemmArgs)") == "1:gw96x16u(1x4)+4"
    assert prof_key("void s2vt::(anonymous namespace)::lstm_bwd_chain4_kernel<64, 5>(s2vt::(anonymous namespace)::BwdChainArgs)") == "6:bchain4(ng64,m320)"
    assert prof_key("void s2vt::(anonymous namespace)::attn_chain_kernel<64>(x)") == "9:attn_chain(ng64)"
    # a stamped summary of another build is not this build's: the lookup says None rather than lend a stale figure
    t, src = bench.stored_traffic(3, "no-such-tile", "rl")
    assert t is None and src is None


def test_python_dash_m_runs_the_drivers_through_the_import_alias():
    """`python -m s2vt_amd.train_rl` (INTEGRATION.md's command lines): runpy asks the alias loader for the module's code; --help needs no GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod, flag in (("train_rl", "--attr-vocab"), ("train_xe", "--train-sents"), ("train_e2e", "--reinforce"), ("train_attention", "--frames")):
        r = subprocess.run([sys.executable, "-m", "s2vt_amd." + mod, "--help"], cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and flag in r.stdout, (mod, r.stderr[-1500:])


# ---- round 6 ---------------------------------------------------------------------------------------------------------------------
def test_multitask_class_surface_constructor_and_tuples():
    import pytest
    """SURVEY 8(b): the multitask / end-to-end class's constructor (reinforce_multitask_e2e_attribute_loss.py:71-76: the base twelve, then
    width, height, channels, feature_dim, label_dim, alpha -- positionally) and the lengths / placeholder shapes of its build_* tuples
    (:226 6-tuple, :380 6-tuple; multitask_e2e_attribute_s2vt.py:223 7-tuple; reinforce_multitask_e2e_attribute_s2vt.py:226,375 5 / 4)."""
    from s2vt_amd import model as M, multitask
    m = M.Video_Caption_Generator(24, 50, 8, 16, 2, 8, 3, 5, None, 1, 0.00005, 0.9, 9, 11, 3, 24, 6, 0.3, device="cpu")
    assert (m.width, m.height, m.channels, m.feature_dim, m.label_dim, m.alpha) == (9, 11, 3, 24, 6, 0.3)
    assert tuple(m.store.p["attr_W"].shape) == (24, 6) and m._decay_everything()
    with pytest.raises(ValueError, match="feature_dim"):
        M.Video_Caption_Generator(24, 50, 8, 16, 2, 8, 3, 5, feature_dim=32, label_dim=6, device="cpu")
    base = M.Video_Caption_Generator(24, 50, 8, 16, 2, 8, 3, 5, device="cpu")
    assert len(base.build_model()) == 5 and len(base.build_loss()) == 4 and not base._decay_everything()
    mm = multitask.Video_Caption_Generator(1536, 50, 8, 16, 2, 8, 3, 5, device="cpu")
    assert (mm.width, mm.height, mm.channels, mm.feature_dim, mm.label_dim, mm.alpha, mm.multisample) == (299, 299, 3, 1536, 400, 0.2, 1)
    loss, video, caption, caption_mask, probs, true_labels = mm.build_model()
    assert video.shape == (2, 3, 1536) and true_labels.shape == (2, 400) and caption.shape == (2, 5)
    assert len(mm.build_model(with_multilabel_loss=True)) == 7
    loss, video, caption, caption_mask, true_labels, multilabel_loss = mm.build_loss()
    assert true_labels.shape == (2, 400) and multilabel_loss.inputs == [video, true_labels]
    # the lambda-mixed script's class: attribute head commented out (:58-60) -> label_dim=0 -> the 5- / 4-tuples, every variable decayed
    s2 = multitask.Video_Caption_Generator(1536, 50, 8, 16, 2, 8, 3, 5, label_dim=0, device="cpu")
    assert len(s2.build_model()) == 5 and len(s2.build_loss()) == 4 and s2._decay_everything()
    # with a CNN attached the placeholders are the frame placeholders [batch, Tv, height, width, channels] (:118)
    s2.e2e = object()
    assert s2.build_model()[1].shape == (2, 3, 299, 299, 3) and s2.build_sampler()[1].shape == (None, 3, 299, 299, 3)
    # train-op wiring: which placeholders each form reads
    s2.e2e = None
    r, b = mm.placeholder("rewards", [None]), mm.placeholder("base_line", [None])
    bl = mm.build_loss()
    op, sl = mm.multitask_train_op(bl, r, b, 1e-3, alpha=0.2)
    assert op.inputs == [bl[1], bl[2], bl[3], r, b, bl[4]] and sl.fn is op.fn
    bm, bl = s2.build_model(), s2.build_loss()
    op, _ = s2.multitask_train_op(bl, r, b, 1e-3, clip_norm=5, build_model_outputs=bm, lambda_loss=0.5)
    assert op.inputs == [bl[1], bl[2], bl[3], r, b, bm[1], bm[2], bm[3]]
    with pytest.raises(ValueError, match="build_model_outputs"):
        s2.multitask_train_op(bl, r, b, 1e-3, lambda_loss=0.5)


def test_profiler_keys_of_the_round6_kernel_names():
    """tools/prof_keys.py maps rocprofv3 kernel names onto the launch profiler's (class : tile) keys that bench.py looks the stamped traffic / SQ
    summaries up with: the LDS-DMA ring tiles carry a twelfth template argument (ring stages), the persistent decode loop files under class 2."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("prof_keys", os.path.join(root, "tools", "prof_keys.py"))
    pk = importlib.util.module_from_spec(spec); spec.loader.exec_module(pk)
    k = pk.prof_key
    assert k("void s2vt::gemm_kernel<2, 2, 4, 4, 1, 0, true, 32, 4, false, false, 2>(s2vt::GemmArgs)") == "0:128x128(2x2)+4dma2"
    assert k("void s2vt::gemm_kernel<2, 2, 3, 3, 1, 0, true, 32, 4, true, false, 2>(s2vt::GemmArgs)") == "4:nt96x96(2x2)+4dma2"
    assert k("void s2vt::gemm_kernel<1, 4, 6, 1, 4, 3, true, 32, 4, false, false, 6>(s2vt::GemmArgs)") == "1:gw96x16u(1x4)+4dma6"
    assert k("void s2vt::gemm_kernel<2, 2, 2, 3, 1, 2, true, 32, 0, false, false, 0>(s2vt::GemmArgs)") == "2:64x96(2x2)"
    assert k("void s2vt::gemm_kernel<1, 4, 1, 1, 4, 3, true, 64, 0, false, true, 0>(s2vt::GemmArgs)") == "1:gw16x16u(1x4)k64[live]"
    assert k("void s2vt::(anonymous namespace)::decode_loop_kernel<1>(s2vt::(anonymous namespace)::DecLoopArgs)") == "2:decloop(m64)"
    assert k("void s2vt::(anonymous namespace)::decode_loop_kernel<6>(s2vt::(anonymous namespace)::DecLoopArgs)") == "2:decloop(m384)"
    assert k("void s2vt::gemm_tn_dma_kernel<false, 16, 2>(s2vt::TnKArgs)") == "3:tn128x128(dma)"
