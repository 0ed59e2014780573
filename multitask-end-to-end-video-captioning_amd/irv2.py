"""Inception-ResNet-v2 feature extractor for the end-to-end scripts (e2e_tf_s2vt.py:106-121):
frames [n, 3, 299, 299] in [-1, 1] -> 1536-d global-average-pooled features, the `net` the reference reshapes to
video[B, Tv, 1536].

The architecture follows the slim definition the reference vendors (inception_resnet_v2.py:31-259: stem, Mixed_5b,
10 x block35(0.17), Mixed_6a, 20 x block17(0.10), Mixed_7a, 9 x block8(0.20), block8 without activation,
Conv2d_7b_1x1) as plain PyTorch modules: on MI355X the convolutions run through MIOpen (PyTorch-ROCm); no hand-written
convolution kernels -- the CNN is not on the path this repository hand-optimises, it feeds it.  As in the reference
the network runs with `is_training=False` batch norm (moving statistics, e2e_tf_s2vt.py:115-116) while its weights
stay trainable.  slim's batch_norm has no gamma (scale=False) and epsilon 0.001.

`load_slim_checkpoint` maps a {tf variable name: ndarray} dump of the slim checkpoint (names
'InceptionResnetV2/<scope>/weights', '.../BatchNorm/{beta,moving_mean,moving_variance}', '.../biases') onto the
modules; TF kernels are HWIO, PyTorch's OIHW.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _pair(k):
    return (k, k) if isinstance(k, int) else tuple(k)


class Unit(nn.Module):
    """slim.conv2d under the IRv2 arg_scope: conv (no bias) + batch_norm(center only, eps 1e-3) + ReLU; or, with
    plain=True (normalizer_fn=None, activation_fn=None), a biased linear convolution (the residual `up` projections)."""

    def __init__(self, scope, cin, cout, k, stride=1, valid=False, plain=False):
        super().__init__()
        kh, kw = _pair(k)
        pad = (0, 0) if valid else (kh // 2, kw // 2)          # SAME with stride 1 and odd kernels is symmetric
        self.scope, self.plain = scope, plain
        self.conv = nn.Conv2d(cin, cout, (kh, kw), stride=stride, padding=pad, bias=plain)
        if not plain:
            self.bn = nn.BatchNorm2d(cout, eps=1e-3, affine=True)
            self.bn.weight.requires_grad_(False)               # slim: scale=False -> gamma fixed at 1
            nn.init.ones_(self.bn.weight)

    def forward(self, x):
        x = self.conv(x)
        if self.plain:
            return x
        # inference-mode statistics whatever the module's train() state (is_training=False in the reference)
        x = F.batch_norm(x, self.bn.running_mean, self.bn.running_var, self.bn.weight, self.bn.bias, False, 0.0, self.bn.eps)
        return F.relu(x)


class Block35(nn.Module):
    def __init__(self, scope, scale):
        super().__init__()
        self.scale = scale
        self.b0 = Unit(f"{scope}/Branch_0/Conv2d_1x1", 320, 32, 1)
        self.b1 = nn.Sequential(Unit(f"{scope}/Branch_1/Conv2d_0a_1x1", 320, 32, 1), Unit(f"{scope}/Branch_1/Conv2d_0b_3x3", 32, 32, 3))
        self.b2 = nn.Sequential(Unit(f"{scope}/Branch_2/Conv2d_0a_1x1", 320, 32, 1), Unit(f"{scope}/Branch_2/Conv2d_0b_3x3", 32, 48, 3),
                                Unit(f"{scope}/Branch_2/Conv2d_0c_3x3", 48, 64, 3))
        self.up = Unit(f"{scope}/Conv2d_1x1", 128, 320, 1, plain=True)

    def forward(self, x):
        return F.relu(x + self.scale * self.up(torch.cat([self.b0(x), self.b1(x), self.b2(x)], 1)))


class Block17(nn.Module):
    def __init__(self, scope, scale):
        super().__init__()
        self.scale = scale
        self.b0 = Unit(f"{scope}/Branch_0/Conv2d_1x1", 1088, 192, 1)
        self.b1 = nn.Sequential(Unit(f"{scope}/Branch_1/Conv2d_0a_1x1", 1088, 128, 1), Unit(f"{scope}/Branch_1/Conv2d_0b_1x7", 128, 160, (1, 7)),
                                Unit(f"{scope}/Branch_1/Conv2d_0c_7x1", 160, 192, (7, 1)))
        self.up = Unit(f"{scope}/Conv2d_1x1", 384, 1088, 1, plain=True)

    def forward(self, x):
        return F.relu(x + self.scale * self.up(torch.cat([self.b0(x), self.b1(x)], 1)))


class Block8(nn.Module):
    def __init__(self, scope, scale, activation=True):
        super().__init__()
        self.scale, self.activation = scale, activation
        self.b0 = Unit(f"{scope}/Branch_0/Conv2d_1x1", 2080, 192, 1)
        self.b1 = nn.Sequential(Unit(f"{scope}/Branch_1/Conv2d_0a_1x1", 2080, 192, 1), Unit(f"{scope}/Branch_1/Conv2d_0b_1x3", 192, 224, (1, 3)),
                                Unit(f"{scope}/Branch_1/Conv2d_0c_3x1", 224, 256, (3, 1)))
        self.up = Unit(f"{scope}/Conv2d_1x1", 448, 2080, 1, plain=True)

    def forward(self, x):
        y = x + self.scale * self.up(torch.cat([self.b0(x), self.b1(x)], 1))
        return F.relu(y) if self.activation else y


class InceptionResnetV2(nn.Module):
    """inception_resnet_v2_base(final_endpoint='Conv2d_7b_1x1') + global average pool: [n,3,H,W] -> [n,1536]."""

    def __init__(self):
        super().__init__()
        R = "InceptionResnetV2"
        self.stem = nn.Sequential(
            Unit(f"{R}/Conv2d_1a_3x3", 3, 32, 3, stride=2, valid=True), Unit(f"{R}/Conv2d_2a_3x3", 32, 32, 3, valid=True),
            Unit(f"{R}/Conv2d_2b_3x3", 32, 64, 3), nn.MaxPool2d(3, 2),
            Unit(f"{R}/Conv2d_3b_1x1", 64, 80, 1, valid=True), Unit(f"{R}/Conv2d_4a_3x3", 80, 192, 3, valid=True), nn.MaxPool2d(3, 2))
        m = f"{R}/Mixed_5b"
        self.m5_b0 = Unit(f"{m}/Branch_0/Conv2d_1x1", 192, 96, 1)
        self.m5_b1 = nn.Sequential(Unit(f"{m}/Branch_1/Conv2d_0a_1x1", 192, 48, 1), Unit(f"{m}/Branch_1/Conv2d_0b_5x5", 48, 64, 5))
        self.m5_b2 = nn.Sequential(Unit(f"{m}/Branch_2/Conv2d_0a_1x1", 192, 64, 1), Unit(f"{m}/Branch_2/Conv2d_0b_3x3", 64, 96, 3),
                                   Unit(f"{m}/Branch_2/Conv2d_0c_3x3", 96, 96, 3))
        self.m5_b3 = Unit(f"{m}/Branch_3/Conv2d_0b_1x1", 192, 64, 1)
        self.repeat = nn.Sequential(*[Block35(f"{R}/Repeat/block35_{i + 1}", 0.17) for i in range(10)])
        m = f"{R}/Mixed_6a"
        self.m6_b0 = Unit(f"{m}/Branch_0/Conv2d_1a_3x3", 320, 384, 3, stride=2, valid=True)
        self.m6_b1 = nn.Sequential(Unit(f"{m}/Branch_1/Conv2d_0a_1x1", 320, 256, 1), Unit(f"{m}/Branch_1/Conv2d_0b_3x3", 256, 256, 3),
                                   Unit(f"{m}/Branch_1/Conv2d_1a_3x3", 256, 384, 3, stride=2, valid=True))
        self.repeat_1 = nn.Sequential(*[Block17(f"{R}/Repeat_1/block17_{i + 1}", 0.10) for i in range(20)])
        m = f"{R}/Mixed_7a"
        self.m7_b0 = nn.Sequential(Unit(f"{m}/Branch_0/Conv2d_0a_1x1", 1088, 256, 1), Unit(f"{m}/Branch_0/Conv2d_1a_3x3", 256, 384, 3, stride=2, valid=True))
        self.m7_b1 = nn.Sequential(Unit(f"{m}/Branch_1/Conv2d_0a_1x1", 1088, 256, 1), Unit(f"{m}/Branch_1/Conv2d_1a_3x3", 256, 288, 3, stride=2, valid=True))
        self.m7_b2 = nn.Sequential(Unit(f"{m}/Branch_2/Conv2d_0a_1x1", 1088, 256, 1), Unit(f"{m}/Branch_2/Conv2d_0b_3x3", 256, 288, 3),
                                   Unit(f"{m}/Branch_2/Conv2d_1a_3x3", 288, 320, 3, stride=2, valid=True))
        self.repeat_2 = nn.Sequential(*[Block8(f"{R}/Repeat_2/block8_{i + 1}", 0.20) for i in range(9)])
        self.block8 = Block8(f"{R}/Block8", 1.0, activation=False)
        self.conv7b = Unit(f"{R}/Conv2d_7b_1x1", 2080, 1536, 1)

    def features(self, x):
        x = self.stem(x)
        x = torch.cat([self.m5_b0(x), self.m5_b1(x), self.m5_b2(x), self.m5_b3(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))], 1)   # Mixed_5b: 320
        x = self.repeat(x)
        x = torch.cat([self.m6_b0(x), self.m6_b1(x), F.max_pool2d(x, 3, 2)], 1)                                  # Mixed_6a: 1088
        x = self.repeat_1(x)
        x = torch.cat([self.m7_b0(x), self.m7_b1(x), self.m7_b2(x), F.max_pool2d(x, 3, 2)], 1)                   # Mixed_7a: 2080
        x = self.block8(self.repeat_2(x))
        return self.conv7b(x)

    def forward(self, frames):
        return self.features(frames).mean(dim=(2, 3))                 # AvgPool_1a_8x8 + flatten (e2e_tf_s2vt.py:118-119)

    def units(self):
        return [m for m in self.modules() if isinstance(m, Unit)]

    def load_slim_checkpoint(self, variables: dict):
        """variables: {tf name: ndarray}.  Returns the names that were loaded."""
        loaded = []
        with torch.no_grad():
            for u in self.units():
                w = variables.get(u.scope + "/weights")
                if w is not None and tuple(w.shape) == tuple(u.conv.weight.permute(2, 3, 1, 0).shape):
                    u.conv.weight.copy_(torch.as_tensor(np.asarray(w, np.float32)).permute(3, 2, 0, 1))         # HWIO -> OIHW
                    loaded.append(u.scope + "/weights")
                if u.plain:
                    b = variables.get(u.scope + "/biases")
                    if b is not None:
                        u.conv.bias.copy_(torch.as_tensor(np.asarray(b, np.float32))); loaded.append(u.scope + "/biases")
                else:
                    for tf_name, t in (("beta", u.bn.bias), ("moving_mean", u.bn.running_mean), ("moving_variance", u.bn.running_var)):
                        a = variables.get(f"{u.scope}/BatchNorm/{tf_name}")
                        if a is not None:
                            t.copy_(torch.as_tensor(np.asarray(a, np.float32))); loaded.append(f"{u.scope}/BatchNorm/{tf_name}")
        return loaded


def preprocess_frames(frames_uint8):
    """image_reading_processing (e2e_tf_s2vt.py:436-447) after decoding/resizing: RGB uint8 [.., H, W, 3] -> float
    [.., 3, H, W] scaled to [-1, 1]."""
    x = torch.as_tensor(frames_uint8).to(torch.float32)
    x = 2.0 * (x / 255.0) - 1.0
    return x.movedim(-1, -3).contiguous()
