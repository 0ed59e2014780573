"""Opt-in GPU run of the end-to-end path with the real Inception-ResNet-v2 (BASELINE configs[4]: batch 16 x 5 frames
x 299 x 299, dim_image 1536, vocab 12000): one XE step and one REINFORCE step through CNN + captioner, timings per
stage.  python tools/e2e_irv2_smoke.py [--batch 16] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import s2vt_amd  # noqa: E402
from s2vt_amd import e2e, irv2, model as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--size", type=int, default=299)
    ap.add_argument("--out", default="")
    ap.add_argument("--channels-last", action="store_true")
    a = ap.parse_args()
    B, Tv, Tc, V = a.batch, 5, 20, 12000
    torch.manual_seed(0)
    mdl = M.Video_Caption_Generator(1536, V, 500, 1000, B, Tv + Tc, Tv, Tc, dropout_rate=0.9)
    net = irv2.InceptionResnetV2()
    tr = e2e.EndToEnd(mdl, net, channels_last=a.channels_last)
    rng = np.random.default_rng(0)
    frames = torch.as_tensor(rng.uniform(-1, 1, (B, Tv, 3, a.size, a.size)).astype(np.float32)).cuda()
    cap = rng.integers(2, V, (B, Tc)).astype(np.int32); cap[:, 12:] = 0
    mask = s2vt_amd.hostglue.masks_from_ids(cap)

    def reward_fn(s, g):
        return rng.random(s.shape[0]).astype(np.float32), rng.random(g.shape[0]).astype(np.float32)
    res = {"batch": B, "frames_per_step": B * Tv, "cnn_params": int(tr.theta.numel())}
    for name, fn in (("xe_step", lambda: tr.xe_step(frames, cap, mask, lr=1e-5)),
                     ("reinforce_step", lambda: tr.reinforce_step(frames, reward_fn, lr=1e-6, K=1, sample_seed=mdl.global_step))):
        t0 = time.time(); st = fn(); torch.cuda.synchronize(); first = time.time() - t0
        ts = []
        for _ in range(a.steps):
            torch.cuda.synchronize(); t0 = time.time(); st = fn(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        res[name] = {"first_s": round(first, 2), "ms_per_step": round(1e3 * float(np.median(ts)), 2), "loss": float(st.loss),
                     "grad_norm": float(st.grad_sumsq.sqrt())}
        assert np.isfinite(res[name]["loss"]) and np.isfinite(res[name]["grad_norm"])
    # stage split of one XE step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); video, h = tr.extract(frames, dropout=True, track=True); ev[1].record()
    h.backward(torch.ones_like(h)); ev[2].record(); torch.cuda.synchronize()
    res["cnn_fwd_ms"] = round(ev[0].elapsed_time(ev[1]), 2); res["cnn_bwd_ms"] = round(ev[1].elapsed_time(ev[2]), 2)
    res["mem_GB"] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
