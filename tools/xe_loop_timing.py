"""Dev helper: wall time per step of the XE training loop (train_xe.train) at the reference's dimensions (B=64, Tc=35)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import s2vt_amd
from s2vt_amd import data, train_common as tc, train_xe

rng = np.random.default_rng(0)
V, nvid, refs_per, D, Tv = 12000, 640, 20, 1536, 5
vocab = ["<en_unk>"] + [f"w{i}" for i in range(V - 3)]
p = 1 / np.arange(1, len(vocab) + 1); p /= p.sum()
sents = [(f"vid{v}", " ".join(vocab[i] for i in rng.choice(len(vocab), size=rng.integers(4, 12), p=p))) for v in range(nvid) for _ in range(refs_per)]
corpus = tc.Corpus.__new__(tc.Corpus)
corpus.captions = np.asarray(sents)
corpus.features = data.FeatureStore(np.abs(rng.standard_normal((nvid, Tv, D)) * 0.5).astype(np.float32), [f"vid{v}" for v in range(nvid)])
corpus.vocabulary = vocab
corpus.index = data.CaptionIndex(corpus.captions)
cfg = tc.Config(n_epochs=1, batch_size=64, max_steps_per_epoch=int(sys.argv[1]) if len(sys.argv) > 1 else 30, model_path="/tmp/xe_timing", step_log="/tmp/xe_timing_steps.jsonl")
if os.path.exists("/tmp/xe_timing_steps.jsonl"):
    os.remove("/tmp/xe_timing_steps.jsonl")
times = []


def log(msg):
    if "Elapsed time" in msg:
        times.append(float(msg.rsplit(":", 1)[1]))


train_xe.train(cfg, corpus, None, log=log)
import json
secs = [json.loads(l)["seconds"] for l in open("/tmp/xe_timing_steps.jsonl") if '"kind": "step"' in l]
print(f"step log: median {1e3 * np.median(secs[3:]):.3f} ms, p10 {1e3 * np.percentile(secs[3:], 10):.3f}, p90 {1e3 * np.percentile(secs[3:], 90):.3f}")
print(f"{len(times)} steps; median wall per XE step {1e3 * np.median(times[3:]):.2f} ms at B=64, Tc=35 -> {64 / np.median(times[3:]):.0f} captions/s")
