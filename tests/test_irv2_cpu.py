"""Inception-ResNet-v2 feature extractor (CPU): stage shapes of the slim definition at 299x299, 1536-d pooled
output, parameter inventory (every slim conv scope present once, HWIO->OIHW load round trip)."""
import numpy as np
import torch


def test_irv2_shapes_and_slim_names():
    import s2vt_amd
    from s2vt_amd import irv2
    torch.manual_seed(0)
    net = irv2.InceptionResnetV2()
    units = net.units()
    scopes = [u.scope for u in units]
    assert len(scopes) == len(set(scopes))
    # 7 stem/Mixed_5b.. counts: stem 5, Mixed_5b 7, 10 x block35 (7), Mixed_6a 4, 20 x block17 (5), Mixed_7a 7, 9 x block8 (5), Block8 5, Conv2d_7b 1
    assert len(units) == 5 + 7 + 10 * 7 + 4 + 20 * 5 + 7 + 9 * 5 + 5 + 1
    assert "InceptionResnetV2/Repeat_1/block17_20/Branch_1/Conv2d_0c_7x1" in scopes and "InceptionResnetV2/Block8/Conv2d_1x1" in scopes
    n_conv = sum(u.conv.weight.numel() for u in units)
    assert 54_000_000 < n_conv < 55_000_000                       # the well-known ~54.3 M convolution weights of the base network
    x = torch.randn(1, 3, 299, 299)
    with torch.no_grad():
        s = net.stem(x)
        assert s.shape == (1, 192, 35, 35)
        f = net.features(x)
        assert f.shape == (1, 1536, 8, 8)
        y = net(x)
    assert y.shape == (1, 1536) and torch.isfinite(y).all() and float(y.min()) >= 0.0          # post-ReLU pooled features
    # slim checkpoint mapping: a fake {name: HWIO array} round-trips into the OIHW parameters
    u = units[3]
    w = np.random.default_rng(0).standard_normal(tuple(u.conv.weight.permute(2, 3, 1, 0).shape)).astype(np.float32)
    loaded = net.load_slim_checkpoint({u.scope + "/weights": w, u.scope + "/BatchNorm/beta": np.ones(u.conv.weight.shape[0], np.float32)})
    assert len(loaded) == 2 and np.array_equal(u.conv.weight.detach().permute(2, 3, 1, 0).numpy(), w)
    assert irv2.preprocess_frames(np.full((2, 4, 4, 3), 255, np.uint8)).shape == (2, 3, 4, 4)


# ---------------------------------------------------------------------------------------------------------------------
# Known-answer tests against the reference's own layer table (tests/golden/irv2_table.json: an ast walk of the reference's
# inception_resnet_v2.py:30-259 made by tools/make_irv2_table.py) and against TF's padding / batch-norm definitions.
# ---------------------------------------------------------------------------------------------------------------------
def _table():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "irv2_table.json")))


def _expected_units(t):
    """Expand the table to {full scope: (out, (kh, kw), stride, valid, plain)} with slim.repeat's block numbering."""
    exp = {}

    def add(e, prefix, in_width):
        k = e["kernel"]
        kh, kw = (k, k) if isinstance(k, int) else k
        out = in_width if e["out"] == "$input_channels" else e["out"]
        exp[prefix + e["scope"].split("/", 1)[1] if prefix else e["scope"]] = (out, (kh, kw), e["stride"], e["padding"] == "VALID", e["plain"])
    repeats = iter(["Repeat", "Repeat_1", "Repeat_2"])
    widths = {"block35": 320, "block17": 1088, "block8": 2080}
    for e in t["base"]:
        if e["op"] == "conv":
            add(e, "", None)
        elif e["op"] == "repeat":
            rep = next(repeats)
            for i in range(e["count"]):
                for be in t[e["block"]]:
                    add(be, f"InceptionResnetV2/{rep}/{e['block']}_{i + 1}/", widths[e["block"]])
        elif e["op"] == "block8_final":
            for be in t["block8"]:
                add(be, "InceptionResnetV2/Block8/", widths["block8"])
    return exp


def test_irv2_units_match_the_reference_layer_table():
    import s2vt_amd
    from s2vt_amd import irv2
    t = _table()
    exp = _expected_units(t)
    net = irv2.InceptionResnetV2()
    got = {u.scope: (u.conv.out_channels, tuple(u.conv.kernel_size), u.conv.stride[0], u.plain) for u in net.units()}
    assert set(got) == set(exp)
    for scope, e in exp.items():
        assert got[scope] == (e[0], e[1], e[2], e[4]), (scope, got[scope], e)
        u = next(u for u in net.units() if u.scope == scope)
        kh, kw = e[1]
        if e[3]:                                                   # VALID: no padding
            assert tuple(u.conv.padding) == (0, 0), scope
        else:                                                      # SAME at stride 1, odd kernel: (k-1)/2 each side
            assert e[2] == 1 and tuple(u.conv.padding) == ((kh - 1) // 2, (kw - 1) // 2), scope
    # every stride-2 op of the table is VALID (so the symmetric-padding Unit never has to emulate TF's asymmetric SAME)
    assert all(e["padding"] == "VALID" for e in t["base"] if e.get("stride") == 2)
    # repeat counts / scales and the final linear block8
    reps = [e for e in t["base"] if e["op"] == "repeat"]
    assert [(r["count"], r["scale"]) for r in reps] == [(10, 0.17), (20, 0.10), (9, 0.20)]
    assert [b.scale for b in net.repeat] == [0.17] * 10 and [b.scale for b in net.repeat_1] == [0.10] * 20
    assert [b.scale for b in net.repeat_2] == [0.20] * 9 and net.block8.scale == 1.0 and not net.block8.activation
    assert [e for e in t["base"] if e["op"] == "block8_final"][0]["activation"] is None


def test_irv2_stage_shapes_match_the_reference_comments():
    """The `# H x W x C` comments of inception_resnet_v2_base at a 299 x 299 input, stage by stage."""
    from s2vt_amd import irv2
    import torch.nn.functional as F
    t = _table()
    want = [tuple(x) for x in t["stage_shape_comments"]]
    net = irv2.InceptionResnetV2()
    x = torch.randn(1, 3, 299, 299)
    shapes = []
    with torch.no_grad():
        for m in net.stem:
            x = m(x)
            shapes.append((x.shape[2], x.shape[3], x.shape[1]))
        x = torch.cat([net.m5_b0(x), net.m5_b1(x), net.m5_b2(x), net.m5_b3(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))], 1)
        shapes.append((x.shape[2], x.shape[3], x.shape[1]))
        x = net.repeat(x)
        x = torch.cat([net.m6_b0(x), net.m6_b1(x), F.max_pool2d(x, 3, 2)], 1)
        shapes.append((x.shape[2], x.shape[3], x.shape[1]))
        x = net.repeat_1(x)
        x = torch.cat([net.m7_b0(x), net.m7_b1(x), net.m7_b2(x), F.max_pool2d(x, 3, 2)], 1)
        shapes.append((x.shape[2], x.shape[3], x.shape[1]))
        x = net.conv7b(net.block8(net.repeat_2(x)))
        shapes.append((x.shape[2], x.shape[3], x.shape[1]))
    assert shapes == want


def _tf_conv2d(x, w, stride, padding):
    """tf.nn.conv2d on NHWC / HWIO in float64 with TF's padding rule: SAME pads total = max((ceil(n/s)-1)*s + k - n, 0),
    the smaller half FIRST; VALID pads nothing."""
    n, H, W, _ = x.shape
    kh, kw, _, co = w.shape
    if padding == "SAME":
        oh, ow = -(-H // stride), -(-W // stride)
        ph, pw = max((oh - 1) * stride + kh - H, 0), max((ow - 1) * stride + kw - W, 0)
        x = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    else:
        oh, ow = (H - kh) // stride + 1, (W - kw) // stride + 1
    out = np.zeros((n, oh, ow, co))
    for i in range(oh):
        for j in range(ow):
            patch = x[:, i * stride:i * stride + kh, j * stride:j * stride + kw, :]
            out[:, i, j, :] = np.tensordot(patch, w, axes=([1, 2, 3], [0, 1, 2]))
    return out


def test_irv2_unit_is_tf_conv_plus_inference_batch_norm():
    """slim.conv2d under the IRv2 arg_scope = conv (no bias) -> (y - moving_mean) / sqrt(moving_variance + 0.001) + beta
    (scale=False) -> ReLU, with TF padding.  One Unit per kernel / stride / padding class that occurs in the table,
    loaded through load_slim_checkpoint's HWIO mapping, against a float64 numpy evaluation."""
    from s2vt_amd import irv2
    rng = np.random.default_rng(0)
    t = _table()
    classes = sorted({(tuple(e["kernel"]) if isinstance(e["kernel"], list) else (e["kernel"], e["kernel"]), e["stride"], e["padding"], e["plain"])
                      for k in ("base", "block35", "block17", "block8") for e in t[k] if e["op"] == "conv"})
    assert ((3, 3), 2, "VALID", False) in classes and ((1, 7), 1, "SAME", False) in classes and ((1, 1), 1, "SAME", True) in classes
    for (kh, kw), stride, padding, plain in classes:
        cin, cout = 5, 6
        u = irv2.Unit("t", cin, cout, (kh, kw), stride=stride, valid=(padding == "VALID"), plain=plain)
        w = rng.standard_normal((kh, kw, cin, cout)).astype(np.float32)
        var = {"t/weights": w}
        if plain:
            var["t/biases"] = rng.standard_normal(cout).astype(np.float32)
        else:
            var.update({"t/BatchNorm/beta": rng.standard_normal(cout).astype(np.float32),
                        "t/BatchNorm/moving_mean": rng.standard_normal(cout).astype(np.float32),
                        "t/BatchNorm/moving_variance": rng.uniform(0.5, 2, cout).astype(np.float32)})
        holder = irv2.InceptionResnetV2.__new__(irv2.InceptionResnetV2)
        torch.nn.Module.__init__(holder)
        holder.u = u
        assert len(holder.load_slim_checkpoint(var)) == len(var)
        x = rng.standard_normal((2, 11, 12, cin)).astype(np.float32)                     # NHWC, odd / even extents
        ref = _tf_conv2d(x.astype(np.float64), w.astype(np.float64), stride, padding)
        if plain:
            ref = ref + var["t/biases"]
        else:
            ref = (ref - var["t/BatchNorm/moving_mean"]) / np.sqrt(var["t/BatchNorm/moving_variance"].astype(np.float64) + 0.001) + var["t/BatchNorm/beta"]
            ref = np.maximum(ref, 0)
        u.train()                                                                        # is_training=False statistics whatever the mode
        with torch.no_grad():
            got = u(torch.as_tensor(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), ((kh, kw), stride, padding, plain)


def test_irv2_pools_follow_tf_padding():
    """AvgPool_0a_3x3 is SAME at stride 1 and TF's average excludes the padding from the divisor; the stride-2 max pools
    are VALID."""
    import torch.nn.functional as F
    x = torch.arange(2 * 1 * 5 * 5, dtype=torch.float32).view(2, 1, 5, 5)
    got = F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)
    ref = torch.zeros_like(x)
    for i in range(5):
        for j in range(5):
            ref[:, :, i, j] = x[:, :, max(i - 1, 0):i + 2, max(j - 1, 0):j + 2].mean(dim=(2, 3))
    assert torch.allclose(got, ref)
    assert F.max_pool2d(torch.randn(1, 1, 35, 35), 3, 2).shape[-1] == 17 and F.max_pool2d(torch.randn(1, 1, 17, 17), 3, 2).shape[-1] == 8
