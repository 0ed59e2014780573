"""dev: the sampler call at the bench shape with samples of realistic lengths (<eos> bias raised), faithful vs early-exit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import model as M, hostglue
B, K = 64, 5
mdl = M.Video_Caption_Generator(1536, 12000, 500, 1000, B, 0, 5, 20, seed=3, multisample=K)
rng = np.random.default_rng(1)
video = torch.as_tensor(np.abs(rng.standard_normal((B, 5, 1536)) * 0.5).astype(np.float32)).cuda()
for bias in (0.0, float(os.environ.get("EOS_BIAS", "7.5"))):
    mdl.store.p["embed_word_b"][0] = bias
    for stop in (False, True):
        for _ in range(3):
            s, g = mdl.sample(video, K, True, seed=5, stop_at_eos=stop)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for i in range(n):
            s, g = mdl.sample(video, K, True, seed=5 + i, stop_at_eos=stop)
        e1.record(); torch.cuda.synchronize()
        ln = hostglue.masks_from_ids(s.cpu().numpy()).sum(1)
        print(f"eos bias {bias:4.1f} stop_at_eos={stop!s:5}  {e0.elapsed_time(e1) / n:7.3f} ms per sampler call   mean length {ln.mean():5.2f}  max {ln.max():.0f}")
