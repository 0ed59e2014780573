"""INTEGRATION.md section 2 is executable: the ctypes stub printed there -- its struct layouts, argtypes and the call itself -- is cut out of
the document and run against libs2vt_hip.so; the token ids it returns equal those of the package's own binding for the same seed."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    (code,) = [b for b in blocks if "L.s2vt_sample.argtypes" in b]
    head, tail = code.split("d = Dims(", 1)
    return head, "d = Dims(" + tail


def test_ctypes_stub_of_integration_md_runs(gpu):
    import torch
    import s2vt_amd
    from s2vt_amd import model as M
    head, tail = _stub()
    B, K, V, seed, video_base = 4, 2, 300, 1234, 3
    mdl = M.Video_Caption_Generator(1536, V, 500, 1000, B, 0, 5, 20, seed=2, multisample=K)
    video = torch.as_tensor(np.abs(np.random.default_rng(0).standard_normal((B, 5, 1536)) * 0.5).astype(np.float32)).cuda()
    want_s, want_g = mdl.sample(video, K, True, seed=seed, video_base=video_base)
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                                              # (the stub opens the library by its path from the repository root)
    try:
        exec(head, ns)                                          # CDLL, Dims, Params, restype / argtypes -- as printed
        p = mdl.store.p
        params = ns["Params"](*[p[n].data_ptr() if n in p else None for n, _ in ns["Params"]._fields_])
        nbytes_probe = ns["L"].s2vt_sample_workspace_bytes(ns["C"].byref(ns["Dims"](1536, V, 500, 1000, 5, 20, 0, 0)), B, K, 1)
        ws = torch.empty(nbytes_probe, dtype=torch.uint8, device="cuda")
        assert ws.data_ptr() % 256 == 0
        ids = torch.full(((K + 1) * B, 20), -1, dtype=torch.int32, device="cuda")
        ns.update(n_words=V, B=B, K=K, seed=seed, video_base=video_base, params=params, video_dev_ptr=video.data_ptr(), ids_dev_ptr=ids.data_ptr(),
                  workspace_dev_ptr=ws.data_ptr(), hip_stream=torch.cuda.current_stream().cuda_stream)
        exec(tail, ns)                                          # d = Dims(...); nbytes = ...; rc = L.s2vt_sample(...)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    assert ns["rc"] == 0 and ns["nbytes"] == nbytes_probe
    assert torch.equal(ids[:K * B], want_s) and torch.equal(ids[K * B:], want_g)
