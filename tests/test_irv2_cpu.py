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
