#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: sampled caption tokens/sec of one
REINFORCE step (BASELINE.json), B=64 per GPU, K=5 samples/video, Tc=20, Tv=5, d=1536, E=500,
H=1000, |V|=12000, fp32, synthetic inputs already resident in HBM.

One step = frame-embed + encode + K multinomial decodes + 1 greedy decode + PG mask (device) +
teacher-forced forward on K*B rows with dropout + reward-scaled NLL + BPTT + (all-reduce) +
global-norm clip + TF-form Adam -- the device work of reinforcement_multisampling_tf_s2vt.py:743-753
and :823-826.  The reward (external CIDEr-D host code) is replaced by synthetic r, b.

  python bench.py --gpus N --steps K --warmup W
N>1: launched by torch.distributed.run, one rank per GPU, RCCL all-reduce (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, K, TC, TV, D, E, H, V = 64, 5, 20, 5, 1536, 500, 1000, 12000
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# SURVEY §8(d) algorithmic flops: F_seq = 7.68 + 5*28 + 20*52 MFLOP, (4K+1)*B sequence-forwards/step
FLOPS_PER_STEP = (7.68e6 + 5 * 28e6 + 20 * 52e6) * (4 * K + 1) * B


def cpu_baseline():
    """The reference-structured step (K sampler passes + greedy pass + fwd/bwd at K*B + clip + Adam,
    reward excluded) on the host cores with torch-CPU fp32 -- oracle/s2vt_torch.py ("port").
    Bounded: a probe (one sampler pass = 1/26 of the step's flops) picks a thread count that is not
    pathological on many-core hosts, and the full step is only run when the probe predicts < 90 s;
    otherwise the probe itself is the sample and the rate is extrapolated by flops."""
    import numpy as np
    import torch
    from oracle import s2vt_oracle as orc
    from oracle import s2vt_torch as T
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

    best = None
    for cores in sorted({min(ncpu, 32), min(ncpu, 8)}, reverse=True):
        torch.set_num_threads(cores)
        probe()                                    # page-in / thread-pool warm-up
        tp = probe()
        if best is None or tp < best[1]:
            best = (cores, tp)
    cores, tp = best
    torch.set_num_threads(cores)
    est = tp * (4 * K + 1)                         # (4K+1) sequence-forward equivalents per step
    if est > 90.0:
        return {"value": K * B * TC / est, "unit": "tokens/s", "cores": cores, "kind": "port",
                "sample": f"one greedy sampler pass at B={B} ({tp:.1f} s, torch-CPU fp32); step rate extrapolated x{4 * K + 1} by flops"}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    nsteps = max(1, min(4, int(15.0 / max(est, 1e-3))))          # ~10-30 s of CPU work
    t0 = time.time()
    for i in range(nsteps):
        T.reference_structured_step(p, m, v, i, video, K, TC, r, b, gen=g)
    dt = (time.time() - t0) / nsteps
    return {"value": K * B * TC / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} full REINFORCE step(s) (B={B}, K={K}, Tc={TC}, |V|={V}) structured as the reference: "
                      f"{K}+1 sampler passes + fwd/bwd at {K * B} rows + clip + Adam, torch-CPU fp32, {cores} threads, {dt:.1f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    local = local % ndev                          # (functional tests may put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("S2VT_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for single-GPU functional tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import s2vt_amd
    from s2vt_amd import model as M
    from s2vt_amd import ops
    s2vt_amd.lib()                                   # no fallback: raises if the HIP library is missing

    mdl = M.Video_Caption_Generator(D, V, E, H, B, 0, TV, TC, device=dev, seed=1234)     # identical replicas
    mdl.world_size, mdl.rank = world, rank
    g = torch.Generator().manual_seed(1234 + rank)                                        # per-rank data shard
    video = (torch.randn(B, TV, D, generator=g) * 0.5).abs().to(dev)                      # post-ReLU IRv2 pool features
    rewards = (torch.rand(K * B, generator=g) * 2).to(dev)
    baseline = (torch.rand(B, generator=g) * 2).repeat(K).to(dev)

    def step(i):
        s, _greedy = mdl.sample(video, K, True, seed=2024 + i, video_base=rank * B)
        is_eos = (s == 0)
        mask = ((torch.cumsum(is_eos.int(), 1) - is_eos.int()) == 0).float()   # 1 up to and incl. first <eos>
        return mdl.reinforce_update(video, s, mask, rewards, baseline, lr=1e-6, clip_norm=5.0, video_base=rank * B,
                                    reuse_sampler_state=True)    # LSTM1 trajectory of the sampler pass (same videos, same weights)

    # Warm-up: every contraction launch is bracketed by HIP events (in-library, on the launching stream) to get the
    # per-kernel table and find the dominant kernel; inside the timed region only THAT kernel keeps its events
    # (two events per launch on all ~600 launches of a step cost ~10 % of the step).
    ops.prof_filter(-1, -1)
    ops.prof_enable(True)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    warm_rows = ops.prof_collect()
    dom_w = max(warm_rows, key=lambda r: r["total_ms"]) if warm_rows else None
    if dom_w:
        ops.prof_filter(dom_w["kernel_class"], dom_w["tile_cfg"])
    else:
        ops.prof_filter(3, 0)          # --warmup 0: no table to pick from; the weight-gradient contraction (tn128x128) is the known dominant kernel
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = step(args.warmup + i)
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

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = K * B * TC * world * args.steps / dt
        dom = max(rows, key=lambda r: r["total_ms"]) if rows else None
        roof = None
        if dom:
            ach = dom["total_flops"] / (dom["total_ms"] * 1e-3) / 1e12
            cls = {0: "contraction+store", 1: "fused LSTM cell (4-gate GEMM + pointwise epilogue)", 2: "vocab logits + Gumbel-max pick",
                   3: "weight-gradient TN contraction", 4: "contraction+store, W^T operand (backward data gradients)"}[dom["kernel_class"]]
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(pmc):
                ent = json.load(open(pmc)).get(f"{dom['kernel_class']}:{dom['name']}")
                traffic = ent["bytes_per_launch"] if ent else None        # HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, tools/collect_pmc.sh)
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "kernel": f"{cls}, tile {dom['name']}", "launches": dom["launches"],
                    "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
                    "share_of_step": round(dom["total_ms"] / (dt * 1e3), 3),
                    "whole_step_frac": round(FLOPS_PER_STEP * args.steps / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "all_kernels_warmup": [{"class": r["kernel_class"], "tile": r["name"], "launches": r["launches"],
                                            "ms": round(r["total_ms"], 2), "tflops": round(r["total_flops"] / (r["total_ms"] * 1e-3) / 1e12, 1)}
                                           for r in sorted(warm_rows, key=lambda r: -r["total_ms"])]}
        out = {"metric": "sampled caption tokens/sec (REINFORCE step)", "value": round(value, 1), "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "reinforcement_multisampling K=5 self-critical REINFORCE step (BASELINE configs[2]): "
                                      "B=64 per GPU, T_vid=5, T_cap=20, d=1536, E=500, H=1000, |V|=12000; synthetic rewards",
                          "global_batch": B * world, "samples_per_video": K, "parallelism": f"dp{world}",
                          "loss": float(st.loss)},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
