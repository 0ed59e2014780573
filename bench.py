#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: sampled caption tokens/sec of one
REINFORCE step (BASELINE.json), B=64 per GPU, K=5 samples/video, Tc=20, Tv=5, d=1536, E=500,
H=1000, |V|=12000, fp32, synthetic inputs already resident in HBM.

One step (default workload "rl", BASELINE configs[2]) = frame-embed + encode + K multinomial decodes + 1 greedy decode
+ PG mask (device) + teacher-forced forward on K*B rows with dropout + reward-scaled NLL + BPTT + (all-reduce) +
global-norm clip + TF-form Adam -- the device work of reinforcement_multisampling_tf_s2vt.py:743-753 and :823-826.  The
reward (external CIDEr-D host code) is replaced by synthetic r, b.

  python bench.py --gpus N --steps K --warmup W [--workload rl|xe|multitask|e2e|e2e_xe]
N>1: one rank per GPU, RCCL all-reduce of the flat gradient bucket (weak scaling: B per GPU fixed).  Either launched by
torch.distributed.run (WORLD_SIZE in the environment), or -- plain `python bench.py --gpus N` -- this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` itself, BEFORE anything here touches the GPU, relays
rank 0's line and exits with the children's code.  The N>1 line proves itself: config.rccl_ranks (world size of the
"nccl" group), config.replica_max_abs_diff (max-min over ranks of checksums of the variables after the timed region:
0.0), config.allreduce_ms (HIP events around the bucket exchange, per step).
Prints ONE JSON line on rank 0.

Other single-GPU configurations of BASELINE.json (their own metric names; the default line is unchanged):
  --workload xe         configs[1]: tf_s2vt.py XE train step, B=64 (label smoothing, Q1, weight decay, clip 10)
  --workload multitask  configs[3] per-GPU shape: B=32 (256 / 8 GPUs), K=1, attribute-FC head (400 labels) + XE mix
                        (lambda 0.5) + REINFORCE (reinforce_multitask_e2e_attribute_s2vt.py:850, ..._loss.py:957)
  --workload e2e        configs[4] per-GPU shape: B=16 (128 / 8 GPUs) x 5 frames x 299^2 through Inception-ResNet-v2
                        (torch / MIOpen fp32) + the captioner, REINFORCE step K=1 (reinforcement_e2e.py:1085-1140);
                        e2e_xe: the XE step of e2e_tf_s2vt.py:482-700.  CNN-bound: the roofline object still describes
                        the dominant kernel of THIS library; `config.cnn_ms` {fwd, bwd} (HIP events around the CNN forward and
                        the backward through it, per step) and `config.cnn_flops_frac` (conv / linear MACs x frames against
                        the fp32 MFMA peak) say what MIOpen took.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TC, TV, D, E, H, V = 20, 5, 1536, 500, 1000, 12000
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# SURVEY §8(d) algorithmic flops per sequence-forward: frame embed 7.68 + 5 encode steps x 28 + 20 decode steps x 52 MFLOP
F_SEQ = 7.68e6 + 5 * 28e6 + 20 * 52e6
# attention captioner (original_attention.py:95-147), per sequence-forward: frame embed to H dims 2 Tv d H + image part 2 Tv H H + per decode
# step {query 2 H H, score + context 4 Tv H, LSTM3 2 (3H)(4H), output layer 2 (3H) H, logits 2 H V}
def f_att_seq(tv):
    return 2.0 * tv * D * H + 2.0 * tv * H * H + TC * (2.0 * H * H + 4.0 * tv * H + 24.0 * H * H + 6.0 * H * H + 2.0 * H * V)


WORKLOADS = {
    # name: (B per GPU, K, sequence-forwards per step, tokens per step, metric, description)
    "rl": dict(B=64, K=5, seqfwd=lambda B, K: (4 * K + 1) * B, tokens=lambda B, K: K * B * TC,
               metric="sampled caption tokens/sec (REINFORCE step)",
               desc="reinforcement_multisampling K=5 self-critical REINFORCE step (BASELINE configs[2]): B=64 per GPU, "
                    "T_vid=5, T_cap=20, d=1536, E=500, H=1000, |V|=12000; synthetic rewards"),
    "rl_msvd": dict(B=64, K=5, seqfwd=lambda B, K: (4 * K + 1) * B, tokens=lambda B, K: K * B * TC,
                    metric="sampled caption tokens/sec (REINFORCE step, samples of MSVD-like lengths)",
                    desc="the rl workload with a stand-in for a TRAINED policy's samples: the sampler runs as in rl (its ids are discarded -- "
                         "a random-initialised model never emits <eos>), the update teacher-forces synthetic captions of MSVD-like lengths "
                         "(1 + min(Poisson(6), 18) words + <eos>, ~40 % of the positions unmasked) with their mask on the host, as train_rl has it: "
                         "padding steps are not unrolled, the vocabulary-sized and LSTM2-gradient products run on the unmasked positions (exact)"),
    "rl_msvd_eos": dict(B=64, K=5, seqfwd=lambda B, K: (4 * K + 1) * B, tokens=lambda B, K: K * B * TC,
                        metric="sampled caption tokens/sec (REINFORCE step, samples of MSVD-like lengths, early-exit sampler)",
                        desc="rl_msvd with a sampler that behaves like a trained policy's as well: the <eos> logit bias is raised (bias_init_vector's job in the "
                             "reference) so that samples end after ~7 words, and the opt-in early-exit sampler (s2vt_sample_ex, S2VT_SAMPLE_STOP_AT_EOS) drops a row "
                             "from the decode loop at its first <eos> (ids up to it bit-identical; the reference samples all T_cap steps and masks afterwards); "
                             "tokens/s still counts K*B*T_cap nominal positions per step"),
    "rl_ref": dict(B=256, K=8, tc=35, v=9972, seqfwd=lambda B, K: (4 * K + 1) * B, tokens=lambda B, K: K * B * 35,
                   metric="sampled caption tokens/sec (REINFORCE step, the reference script's own default configuration)",
                   desc="reinforcement_multisampling_tf_s2vt.py's OWN defaults (:505-517, :743-753): batch 256, K=8 samples per video, T_cap=35, "
                        "|V|=9972 (msvd_vocabulary1.txt + <bos>/<eos>), T_vid=5, d=1536, E=500, H=1000 -- 2304 sampler rows and N=2048 update rows "
                        "(beyond every persistent form: per-step recurrences), 2.9 GB of logits per pass; synthetic rewards"),
    "xe": dict(B=64, K=0, seqfwd=lambda B, K: 3 * B, tokens=lambda B, K: B * TC,
               metric="caption tokens/sec (XE train step)",
               desc="tf_s2vt XE train step (BASELINE configs[1]): B=64, T_vid=5, T_cap=20, d=1536, E=500, H=1000, |V|=12000; "
                    "label smoothing 0.05, Q1 batch-mean CE, weight decay, clip 10; MSVD-like caption lengths"),
    "attention": dict(B=64, K=0, tv=5, seqfwd=lambda B, K: 3 * B, tokens=lambda B, K: B * TC,
                      metric="caption tokens/sec (temporal-attention XE train step)",
                      desc="original_attention.py train step (soft temporal attention over the frame features + LSTM3 + tanh output layer): "
                           "B=64, T_vid=5, T_cap=20, d=1536, H=1000, |V|=12000; dropout 0.9, clip 10, Adam; MSVD-like caption lengths, "
                           "the padding behind the batch's longest caption is not unrolled"),
    "attention32": dict(B=64, K=0, tv=32, seqfwd=lambda B, K: 3 * B, tokens=lambda B, K: B * TC,
                        metric="caption tokens/sec (temporal-attention XE train step, 32 frames)",
                        desc="original_attention.py train step, the script's '32img' model (:287-290): T_vid=32 -- the alpha regulariser "
                             "beta*max(0, m - sum(alpha[:, :8])) is live; B=64, T_cap=20, d=1536, H=1000, |V|=12000"),
    "multitask": dict(B=32, K=1, seqfwd=lambda B, K: 8 * B, tokens=lambda B, K: K * B * TC,
                      metric="sampled caption tokens/sec (multitask REINFORCE step)",
                      desc="multitask attribute-FC + XE mix + REINFORCE step, per-GPU shape of BASELINE configs[3]: B=32 "
                           "(256 / 8 GPUs), K=1, 400 attribute labels, lambda=0.5, alpha=0.05, features precomputed"),
    "e2e": dict(B=16, K=1, seqfwd=lambda B, K: 8 * B, tokens=lambda B, K: K * B * TC,
                metric="sampled caption tokens/sec (end-to-end IRv2 REINFORCE step)",
                desc="end-to-end Inception-ResNet-v2 fine-tune + REINFORCE step, per-GPU shape of BASELINE configs[4]: B=16 "
                     "(128 / 8 GPUs) x 5 frames x 3x299x299, K=1, |V|=12000; CNN on torch/MIOpen fp32, captioner on this library"),
    "e2e_xe": dict(B=16, K=0, seqfwd=lambda B, K: 3 * B, tokens=lambda B, K: B * TC,
                   metric="caption tokens/sec (end-to-end IRv2 XE step)",
                   desc="end-to-end Inception-ResNet-v2 fine-tune, XE step of e2e_tf_s2vt.py (BASELINE configs[4] path): B=16 per GPU "
                        "x 5 frames x 3x299x299, |V|=12000; CNN on torch/MIOpen fp32, captioner on this library"),
}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def launch_ranks(args):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N ranks under torch.distributed.run as a CHILD process and relay
    rank 0's JSON line.  Nothing in this process has touched the GPU (torch is not even imported; a process that has
    initialised HIP must never exec / fork workers on this pool)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--workload", args.workload]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.batch_per_gpu > 0:
        cmd += ["--batch-per-gpu", str(args.batch_per_gpu)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    return r.returncode if (r.returncode or line) else 1


def kernel_signature():
    """Hash of the kernel sources: profiles/*_pmc_traffic.json is stamped with it, and a stamp that does not match the
    sources this run was built from means the stored HBM-traffic figure is stale -> `traffic: null`."""
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "multitask-end-to-end-video-captioning_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(cs, f), "rb").read())
    return h.hexdigest()[:16]


def stored_traffic(kernel_class, name, workload="rl"):
    """HBM bytes per launch of (class:tile) from the newest stamped PMC summary under profiles/ (2*FETCH_SIZE +
    WRITE_SIZE, separate --pmc passes, tools/collect_round.sh), or None when there is none for THIS build of the kernels.
    Every workload has its own summary (<tag>_pmc_traffic_<workload>.json; the default workload's is <tag>_pmc_traffic.json): the
    same tile moves different bytes per launch on another workload's shapes, so a figure is never borrowed across workloads."""
    prof = os.path.join(ROOT, "profiles")
    sig = kernel_signature()
    sfx = "_pmc_traffic.json" if workload == "rl" else f"_pmc_traffic_{workload}.json"
    for f in sorted((x for x in os.listdir(prof) if x.endswith(sfx)), reverse=True):
        try:
            d = json.load(open(os.path.join(prof, f)))
        except Exception:
            continue
        if d.get("_stamp", {}).get("kernel_signature") != sig:
            continue
        ent = d.get(f"{kernel_class}:{name}")
        if ent:
            return ent["bytes_per_launch"], f
    return None, None


def stored_sq(kernel_class, name, workload="rl"):
    """SQ-counter figures of (class:tile) from the newest stamped summary under profiles/ (<tag>_sq_counters[_<workload>].json, written by
    tools/sq_to_json.py from separate rocprofv3 --pmc passes, tools/collect_round.sh) -- or None when there is none for THIS build of the
    kernels.  mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the dispatch's cycles, GRBM_GUI_ACTIVE / 8): the share of the
    chip's matrix-pipe cycles that issued MFMAs, i.e. MFMA utilisation against the peak at the clock the launch ran at."""
    prof = os.path.join(ROOT, "profiles")
    sig = kernel_signature()
    sfx = "_sq_counters.json" if workload == "rl" else f"_sq_counters_{workload}.json"
    for f in sorted((x for x in os.listdir(prof) if x.endswith(sfx)), reverse=True):
        try:
            d = json.load(open(os.path.join(prof, f)))
        except Exception:
            continue
        if d.get("_stamp", {}).get("kernel_signature") != sig:
            continue
        ent = d.get(f"{kernel_class}:{name}")
        if ent:
            return ent, f
    return None, None


PEAK_HBM_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cpu_baseline():
    """The reference-structured step (K sampler passes + greedy pass + fwd/bwd at K*B + clip + Adam,
    reward excluded) on the host cores with torch-CPU fp32 -- oracle/s2vt_torch.py ("port").
    Bounded: a probe (one sampler pass = 1/26 of the step's flops) picks a thread count that is not
    pathological on many-core hosts, and the full step is only run when the probe predicts < 90 s;
    otherwise the probe itself is the sample and the rate is extrapolated by flops."""
    import torch
    from oracle import s2vt_oracle as orc
    from oracle import s2vt_torch as T
    B, K = 64, 5
    d = orc.Dims(D, V, E, H, TV, TC, 0)
    p = T.to_torch(orc.init_params(d, 1234), torch.float32, True)
    g = torch.Generator().manual_seed(1234)
    video = (torch.randn(B, TV, D, generator=g) * 0.5).abs()
    r = torch.rand(K * B, generator=g) * 2
    b = (torch.rand(B, generator=g) * 2).repeat(K)
    ncpu = os.cpu_count() or 1

    def probe():
        t0 = time.time()
        with torch.no_grad():
            T.unroll(p, video, lambda t, lg: torch.ones(B, dtype=torch.long) if t == 0 else lg.argmax(1), TC)
        return time.time() - t0

    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:
        phys = ncpu
    best, tried = None, {}
    for cores in sorted({min(ncpu, c) for c in (8, 16, 32, 64, phys)}, reverse=True):     # the best the host does, not a guess
        torch.set_num_threads(cores)
        probe()                                    # page-in / thread-pool warm-up
        tp = probe()
        tried[cores] = round(tp, 3)
        if best is None or tp < best[1]:
            best = (cores, tp)
    cores, tp = best
    probe_note = {"probe_seconds_by_threads": tried, "physical_cores": phys}
    torch.set_num_threads(cores)
    est = tp * (4 * K + 1)                         # (4K+1) sequence-forward equivalents per step
    if est > 90.0:
        return {"value": K * B * TC / est, "unit": "tokens/s", "cores": cores, "host_cores": ncpu, "cpu_model": cpu_model(), "kind": "port", **probe_note,
                "sample": f"one greedy sampler pass at B={B} ({tp:.1f} s, torch-CPU fp32); step rate extrapolated x{4 * K + 1} by flops"}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    # The 64-row sampler pass is the LEAST parallel leg of the step; the fwd/bwd at K*B = 320 rows takes more threads.  The thread count the
    # baseline runs with is therefore chosen on the WHOLE reference-structured step: one step per candidate (after one untimed step that pages
    # the weights and autograd buffers in), candidates the sampler probe's pick and the larger pool sizes up to the physical cores -- bounded:
    # no further candidates once ~45 s of probing have been spent.
    step_i = [0]

    def one_step():
        t0 = time.time()
        T.reference_structured_step(p, m, v, step_i[0], video, K, TC, r, b, gen=g)
        step_i[0] += 1
        return time.time() - t0
    one_step()
    cand = sorted({min(ncpu, c) for c in (cores, 16, 32, 64, phys) if c >= min(cores, 16)})
    by_step, spent = {}, 0.0
    for c in cand:
        if spent > 45.0 and by_step:
            break
        torch.set_num_threads(c)
        ts = one_step()
        by_step[c] = round(ts, 3)
        spent += ts
    cores_step = min(by_step, key=by_step.get)
    torch.set_num_threads(cores_step)
    probe_note["step_seconds_by_threads"] = by_step
    probe_note["threads_chosen_on"] = "one whole reference-structured step per candidate (sampler-pass probe listed beside it)"
    nsteps = max(1, min(4, int(15.0 / max(by_step[cores_step], 1e-3))))          # ~10-30 s of CPU work
    t0 = time.time()
    for i in range(nsteps):
        one_step()
    dt = (time.time() - t0) / nsteps
    return {"value": K * B * TC / dt, "unit": "tokens/s", "cores": cores_step, "host_cores": ncpu, "cpu_model": cpu_model(), "kind": "port", **probe_note,
            "sample": f"{nsteps} full REINFORCE step(s) (B={B}, K={K}, Tc={TC}, |V|={V}) structured as the reference: "
                      f"{K}+1 sampler passes + fwd/bwd at {K * B} rows + clip + Adam, torch-CPU fp32, {cores_step} threads (the fastest of "
                      f"{sorted(by_step)} on the whole step), {dt:.1f} s/step"}


def make_e2e_step(workload, mdl, dev, rank, B, K, info):
    """configs[4]: frames [B, 5, 3, 299, 299] resident in HBM -> Inception-ResNet-v2 (torch / MIOpen) -> captioner."""
    import numpy as np
    import torch
    from s2vt_amd import e2e, irv2, hostglue
    torch.manual_seed(1234)                                                                # identical CNN replicas
    tr = e2e.EndToEnd(mdl, irv2.InceptionResnetV2(), seed=1234)
    info["cnn_params"] = int(tr.theta.numel())
    info["trainer"] = tr
    info["cnn_macs_per_frame"] = tr.conv_macs_per_frame(299, 299)
    rng = np.random.default_rng(1234 + rank)
    frames = torch.as_tensor(rng.uniform(-1, 1, (B, TV, 3, 299, 299)).astype(np.float32)).to(dev)
    g = torch.Generator().manual_seed(1234 + rank)
    r_dev = (torch.rand(max(K, 1) * B, generator=g) * 2).to(dev)
    b_dev = (torch.rand(B, generator=g) * 2).to(dev)
    if workload == "e2e":
        def step(i):
            return tr.reinforce_step(frames, lambda s_, g_: (r_dev, b_dev), lr=1e-6, K=K, clip_norm=10.0, video_base=rank * B,
                                     sample_seed=2024 + i)
        return step
    ln = 1 + np.minimum(rng.poisson(6, B), TC - 2)
    cap = rng.integers(2, V, (B, TC)).astype(np.int32)
    for j in range(B):
        cap[j, ln[j]:] = 0
    gt = torch.as_tensor(cap).to(dev)
    gt_mask = torch.as_tensor(hostglue.masks_from_ids(cap)).to(dev)

    def step(i):
        return tr.xe_step(frames, gt, gt_mask, lr=1e-5, clip_norm=10.0, video_base=rank * B)
    return step


def make_step(workload, mdl, dev, rank, B, K, info=None):
    """Synthetic inputs, resident in HBM, and the step closure of the workload."""
    import numpy as np
    import torch
    from s2vt_amd import ops
    if workload.startswith("e2e"):
        return make_e2e_step(workload, mdl, dev, rank, B, K, info if info is not None else {})
    g = torch.Generator().manual_seed(1234 + rank)                                        # per-rank data shard
    tv = WORKLOADS[workload].get("tv", TV)
    TC, V = WORKLOADS[workload].get("tc", globals()["TC"]), WORKLOADS[workload].get("v", globals()["V"])
    video = (torch.randn(B, tv, D, generator=g) * 0.5).abs().to(dev)                      # post-ReLU IRv2 pool features
    rewards = (torch.rand(max(K, 1) * B, generator=g) * 2).to(dev)
    baseline = (torch.rand(B, generator=g) * 2).repeat(max(K, 1)).to(dev)

    def pg_mask(s):
        is_eos = (s == 0)
        return ((torch.cumsum(is_eos.int(), 1) - is_eos.int()) == 0).float()   # 1 up to and incl. first <eos>

    if workload in ("rl", "rl_ref"):
        def step(i):
            s, _greedy = mdl.sample(video, K, True, seed=2024 + i, video_base=rank * B)
            return mdl.reinforce_update(video, s, None, rewards, baseline, lr=1e-6, clip_norm=5.0, video_base=rank * B,      # mask None: PG mask from the ids, in the library
                                        reuse_sampler_state=True)    # LSTM1 trajectory of the sampler pass (same videos, same weights)
        return step
    if workload in ("rl_msvd", "rl_msvd_eos"):
        stop = workload == "rl_msvd_eos"
        if stop:
            mdl.store.p["embed_word_b"][0] = 7.5          # P(<eos>) per step ~ 1/7: samples of MSVD-like lengths from random weights
            if info is not None:
                info["sampler"] = "stop_at_eos, <eos> bias 7.5"
        rng = np.random.default_rng(4321 + rank)
        N = K * B
        ln = 1 + np.minimum(rng.poisson(6, N), TC - 2)
        cap = rng.integers(2, V, (N, TC)).astype(np.int32)
        for j in range(N):
            cap[j, ln[j]:] = 0
        mask = (np.arange(TC)[None, :] <= ln[:, None]).astype(np.float32)          # words + the first <eos>, host-resident
        capd = torch.as_tensor(cap).to(dev)
        if info is not None:
            info["active_steps"] = mdl.active_steps(mask)
            info["live_fraction"] = round(float(mask.mean()), 3)

        def step(i):
            mdl.sample(video, K, True, seed=2024 + i, video_base=rank * B, stop_at_eos=stop)
            return mdl.reinforce_update(video, capd, mask, rewards, baseline, lr=1e-6, clip_norm=5.0, video_base=rank * B,
                                        reuse_sampler_state=True)
        return step
    # ground-truth captions: length 1 + min(Poisson(6), Tc - 2) words (MSVD mean 7.03), tokens U{2..V-1}, then <eos> = 0
    rng = np.random.default_rng(1234 + rank)
    ln = 1 + np.minimum(rng.poisson(6, B), TC - 2)
    cap = rng.integers(2, V, (B, TC)).astype(np.int32)
    for j in range(B):
        cap[j, ln[j]:] = 0
    gt = torch.as_tensor(cap).to(dev)
    gt_mask = pg_mask(gt)
    if workload.startswith("attention"):
        mask_host = gt_mask.cpu().numpy()                    # host-resident, as the reference's loop has it (sentence_padding_toix)
        if info is not None:
            info["active_steps"] = mdl._active_steps(mask_host, TC)

        steps_att = mdl._active_steps(mask_host, TC)          # ... the host reads the unroll length off its mask once; the mask itself is resident
                                                               # in HBM like every other input of the timed region (no per-step upload in the trace)
        def step(i):
            return mdl.xe_update(video, gt, gt_mask, lr=1e-4, clip_norm=10.0, video_base=rank * B, active_steps=steps_att)
        return step
    if workload == "xe":
        # as train_xe does: the steps behind the longest caption of the GLOBAL batch (its <eos> included) are padding on
        # every rank and are not unrolled (exact: they add zeros); every rank re-draws the other ranks' lengths
        world = int(os.environ.get("WORLD_SIZE", "1"))
        steps = min(TC, 1 + max(int((1 + np.minimum(np.random.default_rng(1234 + r).poisson(6, B), TC - 2)).max()) for r in range(world)))
        if info is not None:
            info["active_steps"] = steps

        def step(i):
            return mdl.xe_update(video, gt, gt_mask, lr=1e-3, clip_norm=10.0, q1=True, video_base=rank * B, active_steps=steps)
        return step
    labels = (torch.rand(B, mdl.label_dim, generator=g) < 0.02).float().to(dev)           # bag-of-words attribute labels, ~8 of 400 set

    def step(i):
        s, _greedy = mdl.sample(video, K, True, seed=2024 + i, video_base=rank * B)
        return mdl.mixed_update(video, s, ops.caption_mask(s, want_target=False)[0], rewards, baseline, gt, gt_mask, lr=1e-6, lambda_loss=0.5, clip_norm=5.0,
                                video_base=rank * B, true_labels=labels, decay_all=True,     # (the script's always-true decay predicate, :222)
                                reuse_sampler_state=True)                                    # LSTM1 trajectory of the sampler pass, as the rl workload
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)      # ~2 s of GPU work at the default workload (SURVEY §8(d) asks >= 50 timed)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="rl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="functional tests only (several ranks on one GPU): overrides the workload's "
                    "B per GPU; the line then carries config.batch_per_gpu_override and is NOT the BASELINE configuration")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    B, K = wl["B"], wl["K"]
    if args.batch_per_gpu > 0:
        B = args.batch_per_gpu
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                 # before torch / HIP is touched in this process
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} rank(s); "
              "the launcher's world is what runs and what n_gpus reports", file=sys.stderr)

    import torch
    import torch.distributed as dist
    import s2vt_amd
    from s2vt_amd import dist as dp
    from s2vt_amd import model as M
    from s2vt_amd import ops
    backend = os.environ.get("S2VT_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm; "gloo" only for single-GPU functional tests
    if int(os.environ.get("WORLD_SIZE", "1")) > max(torch.cuda.device_count(), 1) and backend == "nccl":
        sys.exit(f"bench.py: {os.environ['WORLD_SIZE']} ranks but {torch.cuda.device_count()} GPU(s) visible: RCCL needs one GPU per rank")
    rank, world, dev = dp.init_from_env()
    s2vt_amd.lib()                                   # no fallback: raises if the HIP library is missing

    multitask = args.workload == "multitask"
    tc_w, v_w = wl.get("tc", TC), wl.get("v", V)
    if args.workload.startswith("attention"):
        from s2vt_amd import attention as A
        mdl = A.Attention_Caption_Generator(D, V, H, B, wl["tv"], TC, 0.9, device=dev, seed=1234)
    else:
        mdl = M.Video_Caption_Generator(D, v_w, E, H, B, 0, TV, tc_w, device=dev, seed=1234, multisample=max(K, 1),
                                        label_dim=400 if multitask else 0, alpha=0.05 if multitask else 0.0)   # identical replicas
    mdl.world_size, mdl.rank = world, rank
    info = {}
    step = make_step(args.workload, mdl, dev, rank, B, K, info)

    def probe_mode(mode, it0):
        """1 untimed + 3 timed steps in one exchange mode, timed as the timed region is (barrier + synchronise, MAX over ranks, so every rank
        reads the same number); the replicas must not drift (0.0 exactly)."""
        mdl.dp_overlap = mode
        step(1000 + it0)                                            # first-use allocations, communicator warm-up
        dist.barrier(); torch.cuda.synchronize()
        t0p = time.perf_counter()
        for j in range(3):
            step(1001 + it0 + j)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tp = torch.tensor([(time.perf_counter() - t0p) / 3 * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        d_ = dp.replica_drift(mdl.store.theta)
        assert d_ == 0.0, f"replicas drifted by {d_} with dp_overlap={mode}"
        return round(float(tp), 3)

    def measure():
        """W warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronise on both sides, MAX over ranks -- in the exchange mode
        mdl.dp_overlap currently names.  Returns what the JSON line is made of."""
        # Warm-up: every contraction launch is bracketed by HIP events (in-library, on the launching stream) to get the
        # per-kernel table and find the dominant kernel; inside the timed region only THAT kernel keeps its events
        # (two events per launch on all launches of a step cost ~10 % of the step).
        # (the process's very first step is kept OUT of the table when there is more than one warm-up step: a kernel's first launch can carry
        #  tens of ms of one-time code loading, and one such launch made the pick the "dominant kernel" of a whole run once)
        nprof = args.warmup - 1 if args.warmup >= 2 else args.warmup
        for i in range(args.warmup - nprof):
            step(i)
        torch.cuda.synchronize()
        ops.prof_filter(-1, -1)
        ops.prof_enable(True)
        for i in range(args.warmup - nprof, args.warmup):
            step(i)
        torch.cuda.synchronize()
        warm_rows = ops.prof_collect()
        dom_w = max(warm_rows, key=lambda r: r["total_ms"]) if warm_rows else None
        if dom_w:
            ops.prof_filter(dom_w["kernel_class"], dom_w["tile_cfg"])
        else:
            ops.prof_filter(3, 6)          # --warmup 0: no table to pick from; the weight-gradient contraction (tn128x128, LDS-DMA form) is the known dominant kernel
        # per-step durations: one event per step boundary on the launching stream (negligible next to ~600 launches)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        dp.timing_enable(world > 1)
        if "trainer" in info:
            info["trainer"].timing_enable(True)          # HIP events around every CNN forward / backward of the timed region (4 per step)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks[0].record()
        st = None
        for i in range(args.steps):
            st = step(args.warmup + i)
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ops.prof_enable(False)
        rows = ops.prof_collect()
        ops.prof_filter(-1, -1)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        ar_ms, ar_n = dp.timing_collect()
        dp.timing_enable(False)
        drift = dp.replica_drift(mdl.store.theta)         # every rank takes part (collective); must be exactly 0.0
        if "trainer" in info:
            drift = max(drift, dp.replica_drift(info["trainer"].theta))
            info["cnn_timing"] = info["trainer"].timing_collect()
            info["trainer"].timing_enable(False)
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        return dict(dt=dt, warm_rows=warm_rows, nprof=nprof, rows=rows, ar_ms=ar_ms, ar_n=ar_n, drift=drift, per_step=per_step, loss=float(st.loss) if st is not None else None,
                    dp_overlap=bool(getattr(mdl, "dp_overlap", False)))

    def make_line(m, probe):
        dt, warm_rows, rows, ar_ms, ar_n, drift, per_step = m["dt"], m["warm_rows"], m["rows"], m["ar_ms"], m["ar_n"], m["drift"], m["per_step"]
        ms_step = dt / args.steps * 1e3
        value = wl["tokens"](B, K) * world * args.steps / dt
        pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3) if per_step else None
        # algorithmic flops of the step: SURVEY 8(d)'s per-sequence figure x the sequence-forwards of the workload -- counted on the
        # decode steps the workload UNROLLS where it skips the padding behind the longest caption (xe, attention: exact zeros, not work),
        # and not stated at all for rl_msvd, whose update runs on the ~40 % unmasked positions only (a dense-step count divided by that
        # step's time is not a fraction of anything: round 3 printed 1.128 there)
        tc_eff = info.get("active_steps", TC) if args.workload in ("xe", "attention", "attention32") else TC
        if "tv" in wl:
            f_seq = 2.0 * wl["tv"] * D * H + 2.0 * wl["tv"] * H * H + tc_eff * (f_att_seq(wl["tv"]) - 2.0 * wl["tv"] * D * H - 2.0 * wl["tv"] * H * H) / TC
        elif "tc" in wl:          # another T_cap / |V|: the decode step's vocabulary term 2 H |V| and the step count follow the workload
            f_seq = 7.68e6 + 5 * 28e6 + tc_w * (28e6 + 2.0 * H * v_w)
        else:
            f_seq = 7.68e6 + 5 * 28e6 + tc_eff * 52e6
        flops_step = f_seq * wl["seqfwd"](B, K)
        dense_defined = args.workload not in ("rl_msvd", "rl_msvd_eos")
        dom = max(rows, key=lambda r: r["total_ms"]) if rows else None
        roof = None
        if dom:
            ach = dom["total_flops"] / (dom["total_ms"] * 1e-3) / 1e12
            cls = {0: "contraction+store", 1: "fused LSTM cell (4-gate GEMM + pointwise epilogue)", 2: "vocab logits + Gumbel-max pick",
                   3: "weight-gradient TN contraction", 4: "contraction+store, W^T operand (backward data gradients)",
                   5: "persistent LSTM recurrence", 6: "persistent LSTM backward recurrence", 7: "attention score + softmax + context",
                   8: "attention backward", 9: "persistent attention recurrence (query, score/softmax/context, LSTM3)",
                   10: "persistent attention backward recurrence"}.get(dom["kernel_class"], "?")
            traffic, traffic_src = stored_traffic(dom["kernel_class"], dom["name"], args.workload)
            # flops the contraction kernels actually executed per step (hoisting, LSTM1 once per video and sampler-state
            # reuse execute fewer than the algorithmic count), from the warm-up table
            executed = sum(r["total_flops"] for r in warm_rows) / max(m["nprof"], 1) if warm_rows else None
            if args.workload == "rl_msvd_eos":
                executed = None        # (the launch profiler prices a live-row launch at its full row count: only the device knows how many rows ran)
            avg_us = dom["total_ms"] * 1e3 / dom["launches"]
            sq, sq_src = stored_sq(dom["kernel_class"], dom["name"], args.workload)
            # north_star: "rocprof HBM GB/s and MFMA utilisation reported against peak" -- both for the dominant kernel, from the stamped
            # counter summaries of THIS build (null when stale): HBM-side bytes per launch / this run's average launch time, and the share
            # of the chip's matrix-pipe cycles that issued MFMAs
            hbm_gbps = round(traffic / (avg_us * 1e-6) / 1e9, 1) if traffic else None
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "hbm_gbps": hbm_gbps, "hbm_peak_gbps": PEAK_HBM_GBPS, "hbm_frac": round(hbm_gbps / PEAK_HBM_GBPS, 4) if hbm_gbps else None,
                    "mfma_busy_pct": sq["mfma_busy_pct"] if sq else None, "mfma_busy_source": sq_src,
                    "kernel": f"{cls}, tile {dom['name']}", "launches": dom["launches"],
                    "avg_launch_us": round(avg_us, 2),
                    "share_of_step": round(dom["total_ms"] / (dt * 1e3), 3),
                    "algorithmic_flops_per_step": flops_step if dense_defined else None,
                    "whole_step_frac": round(flops_step * args.steps / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if dense_defined else None,
                    "executed_flops_per_step": executed,
                    "executed_flops_frac": round(executed * args.steps / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if executed else None,
                    "all_kernels_warmup": [{"class": r["kernel_class"], "tile": r["name"], "launches": r["launches"],
                                            "ms": round(r["total_ms"], 2), "tflops": round(r["total_flops"] / (r["total_ms"] * 1e-3) / 1e12, 1)}
                                           for r in sorted(warm_rows, key=lambda r: -r["total_ms"])]}
        out = {"metric": wl["metric"], "value": round(value, 1), "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "step_ms": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "how": "HIP events per step on the launching stream"},
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": wl["desc"], "global_batch": B * world, "samples_per_video": K, "parallelism": f"dp{world}",
                          "dp_overlap": m["dp_overlap"], "dp_overlap_probe_ms": probe, "loss": m["loss"],
                          "persistent_recurrence_timeouts": ops.chain_timeouts()},      # grid-wide waits that gave up: must be 0
               "roofline": roof}
        if world > 1:
            out["config"].update({
                "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,    # 0 = NOT an RCCL run (functional gloo run)
                "dist_backend": dist.get_backend(),
                "replica_max_abs_diff": drift,                                                   # checksums of the variables, max - min over ranks
                "allreduce_ms": round(ar_ms / max(args.steps, 1), 4),                            # per step, rank 0's stream, HIP events
                "allreduce_brackets_per_step": round(ar_n / max(args.steps, 1), 2),
                "gradient_bucket_bytes": int(mdl.store.grad.numel() * 4)})
            # reading aids for the first real multi-GPU line: the rate the bucket moved at (algorithm bandwidth: bucket bytes / time the
            # step's stream spent on or behind the exchange) and what the step would take with the exchange fully hidden behind the
            # backward (S2VT_DP_OVERLAP=1 is the knob; this is a projection from THIS run's numbers, not a measurement)
            ar_s = ar_ms / max(args.steps, 1) / 1e3
            out["config"].update({
                "allreduce_bytes_per_s": round(mdl.store.grad.numel() * 4 / ar_s, 1) if ar_s > 0 else None,
                "ms_per_step_if_exchange_hidden": round(max(ms_step - ar_s * 1e3, ar_s * 1e3), 3),
                "value_if_exchange_hidden": round(wl["tokens"](B, K) * world / (max(ms_step - ar_s * 1e3, ar_s * 1e3) / 1e3), 1)})
        if "cnn_params" in info:
            out["config"]["cnn_params"] = info["cnn_params"]
            out["config"]["note"] = ("CNN-bound step: the convolutions run on MIOpen through PyTorch (not a kernel of this library); "
                                     "the roofline object describes this library's dominant kernel only, config.cnn_ms / cnn_flops_frac the CNN half")
            ct = info.get("cnn_timing")
            if ct:
                # the CNN half of the step (e2e_tf_s2vt.py:112-121): HIP events around EndToEnd.extract's network call and around the backward
                # through it, per step; priced at 2 flops per multiply-accumulate of the convolution / linear layers x frames, backward = 2 x
                # forward (data + weight gradients), against the fp32 MFMA peak (MIOpen's fp32 convolutions: what the reference's arithmetic is)
                fwd_ms, bwd_ms = ct["fwd"] / args.steps, ct["bwd"] / args.steps
                frames = B * TV
                f_fwd = 2.0 * info["cnn_macs_per_frame"] * frames * ct["fwd_calls"] / args.steps
                f_bwd = 4.0 * info["cnn_macs_per_frame"] * frames * ct["bwd_calls"] / args.steps
                out["config"]["cnn_ms"] = {"fwd": round(fwd_ms, 3), "bwd": round(bwd_ms, 3), "share_of_step": round((fwd_ms + bwd_ms) / ms_step, 3),
                                           "fwd_calls_per_step": ct["fwd_calls"] / args.steps, "bwd_calls_per_step": ct["bwd_calls"] / args.steps,
                                           "how": "HIP events on the launching stream around the CNN forward / the backward through it"}
                out["config"]["cnn_gmacs_per_frame"] = round(info["cnn_macs_per_frame"] / 1e9, 3)
                out["config"]["cnn_flops_frac"] = {"fwd": round(f_fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if fwd_ms else None,
                                                   "bwd": round(f_bwd / (bwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if bwd_ms else None,
                                                   "both": round((f_fwd + f_bwd) / ((fwd_ms + bwd_ms) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if fwd_ms + bwd_ms else None,
                                                   "peak_tflops": PEAK_FP32_MFMA_TFLOPS}
        if args.batch_per_gpu > 0:
            out["config"]["batch_per_gpu_override"] = B
        if "active_steps" in info:
            out["config"]["unrolled_caption_steps"] = info["active_steps"]      # of TC: behind the longest caption all is padding
        if "live_fraction" in info:
            out["config"]["unmasked_positions"] = info["live_fraction"]
        if "sampler" in info:
            out["config"]["sampler"] = info["sampler"]
        if world == 1 and not args.no_cpu_baseline and args.workload == "rl":
            out["cpu_baseline"] = cpu_baseline()
        return out

    # Data parallel: let the run choose between the blocking bucket exchange and the overlapped slices (S2VT_DP_OVERLAP / model.dp_overlap) by
    # itself -- the driver runs one fixed command.  Order: (1) 3 probe steps with the blocking exchange; (2) the whole measurement in that mode,
    # whose line is then READY; (3) behind a watchdog, 3 probe steps with the overlapped slices; (4) if they were faster, the whole measurement
    # again in that mode, and ITS line is the one printed.  The overlapped form has run on hardware with one RCCL rank and with gloo ranks only
    # (no multi-GPU box was available to the build): should it stall on a real multi-rank communicator, the watchdog prints the blocking-mode
    # line and ends the rank -- the run still yields its number.  An explicit S2VT_DP_OVERLAP in the environment is respected and not probed.
    probe_ok = world > 1 and hasattr(mdl, "dp_overlap") and args.workload in ("rl", "rl_ref", "rl_msvd", "rl_msvd_eos", "xe") and "S2VT_DP_OVERLAP" not in os.environ
    if not probe_ok:
        m_ = measure()
        if rank == 0:
            print(json.dumps(make_line(m_, None)), flush=True)
    else:
        import threading
        probe = {"off": probe_mode(False, 0), "on": None}
        mdl.dp_overlap = False
        m_off = measure()
        ms_off = m_off["dt"] / args.steps * 1e3
        line_off = make_line(m_off, dict(probe)) if rank == 0 else None       # (rank 0 only: the line reads profiles/ and, at N=1, times the CPU baseline)
        done = threading.Event()
        # A stalled or failed overlapped mode must not cost the run its number: the blocking-mode line is complete at this point, and every way out of
        # the overlapped attempt below -- the watchdog (a hang), an exception on this rank (replica drift, an RCCL error), the normal end -- prints
        # exactly ONE line on rank 0.  The abnormal ways leave through os._exit without another collective (a barrier could hang on the very
        # communicator that just failed); their exit code is 0 so that the run yields its number, or S2VT_BENCH_STALL_EXIT_CODE for callers that want
        # to tell such a run from a clean one.
        bad_exit = int(os.environ.get("S2VT_BENCH_STALL_EXIT_CODE", "0"))

        def give_up(why):
            if done.is_set():
                return
            done.set()
            if rank == 0:
                line_off["config"]["dp_overlap_probe_ms"] = {"off": probe["off"], "on": f"{why}: the blocking-mode measurement is reported"}
                print(json.dumps(line_off), flush=True)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(bad_exit)
        deadline = float(os.environ.get("S2VT_BENCH_WATCHDOG_S", "0")) or max(120.0, 40.0 * ms_off * args.steps / 1e3)
        wd = threading.Timer(deadline, give_up, args=("stalled",))
        wd.daemon = True
        wd.start()
        out = None
        try:
            if os.environ.get("S2VT_BENCH_FAKE_STALL") == "1":      # test hook (tests/test_gpu_dp.py): the overlapped mode "hangs" -- the watchdog must deliver the line
                time.sleep(1e6)
            if os.environ.get("S2VT_BENCH_FAKE_FAIL") == "1":       # test hook: the overlapped mode raises (what a drifted replica or an RCCL error does)
                raise AssertionError("replicas drifted by 1.0 with dp_overlap=True (S2VT_BENCH_FAKE_FAIL)")
            probe["on"] = probe_mode(True, 100)
            if probe["on"] < probe["off"]:
                mdl.dp_overlap = True
                m_on = measure()
                out = make_line(m_on, dict(probe)) if rank == 0 else None
            else:
                mdl.dp_overlap = False
                out = line_off
                if rank == 0:
                    out["config"]["dp_overlap_probe_ms"] = dict(probe)
        except BaseException as e:                                    # noqa: BLE001 -- whatever it was, the blocking-mode line stands
            print(f"bench.py: rank {rank}: overlapped exchange failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            give_up(f"failed on rank {rank} ({type(e).__name__}: {str(e)[:160]})")
        done.set()
        wd.cancel()
        if rank == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
