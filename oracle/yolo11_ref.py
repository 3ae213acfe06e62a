"""CPU oracle (TEST INFRASTRUCTURE ONLY - never imported by the product path) for the YOLO11 detection network.

PARITY UNPINNED: the reference reaches this network only through the third-party `ultralytics` package
(`YOLO("yolo11*.pt")`, README.md:200-207; scripts/run.py:347), which is absent here and pinned by none of the reference's
tests.  This file restates the public module definitions in plain torch fp32, module by module the way ultralytics
composes them (Conv, Bottleneck, C3k, C3k2, SPPF, Attention, PSABlock, C2PSA, Detect with the depth-wise class branch),
independently of caesar_yolo_amd/yolo11_graph.py (which lowers the same network to channel-slice ops).  Self-consistency
anchor: the parameter counts of the five scales at nc=80 reproduce the published 2.6 / 9.4 / 20.1 / 25.3 / 56.9 M.

Weights come in as {state_dict conv name: (W[co, ci/groups, k, k], b[co])} with BatchNorm already folded.
Post-processing (decode, NMS) is shared with YOLOv8: oracle/yolov8_ref.py.
"""
import math
import torch
import torch.nn.functional as F

SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)}


class Net11(object):
    def __init__(self, weights, scale="n", nc=80):
        self.w = {k: (torch.as_tensor(v[0], dtype=torch.float32), torch.as_tensor(v[1], dtype=torch.float32))
                  for k, v in weights.items()}
        self.scale, self.nc = scale, nc
        d = SCALES[scale][0]
        self.n2 = max(round(2 * d), 1)
        self.c3k_all = scale in "mlx"
        self.taps = None

    # ---- Conv (conv + folded BN + SiLU unless act=False); groups inferred from the weight shape
    def conv(self, name, x, s=1, act=True):
        w, b = self.w[name]
        groups = x.shape[1] // w.shape[1]
        y = F.conv2d(x, w, b, stride=s, padding=w.shape[-1] // 2, groups=groups)
        y = F.silu(y) if act else y
        if self.taps is not None:
            self.taps[name] = y
        return y

    def bottleneck(self, p, x, shortcut=True):
        y = self.conv(p + ".cv2", self.conv(p + ".cv1", x))
        return x + y if shortcut and x.shape[1] == y.shape[1] else y

    def c3k(self, p, x, shortcut=True):                      # C3 with two 3x3/3x3 bottlenecks at half width
        a = self.conv(p + ".cv1", x)
        for q in range(2):
            a = self.bottleneck("%s.m.%d" % (p, q), a, shortcut)
        return self.conv(p + ".cv3", torch.cat((a, self.conv(p + ".cv2", x)), 1))

    def c3k2(self, i, x, c3k, shortcut=True):
        p = "model.%d" % i
        y = list(self.conv(p + ".cv1", x).chunk(2, 1))
        for j in range(self.n2):
            m = "%s.m.%d" % (p, j)
            y.append(self.c3k(m, y[-1], shortcut) if (c3k or self.c3k_all) else self.bottleneck(m, y[-1], shortcut))
        return self.conv(p + ".cv2", torch.cat(y, 1))

    def sppf(self, x):
        a = self.conv("model.9.cv1", x)
        b = F.max_pool2d(a, 5, 1, 2)
        c = F.max_pool2d(b, 5, 1, 2)
        d = F.max_pool2d(c, 5, 1, 2)
        return self.conv("model.9.cv2", torch.cat((a, b, c, d), 1))

    def attention(self, p, x):
        B, C, H, W = x.shape
        N = H * W
        heads = C // 64
        hd = C // heads
        kd = hd // 2
        qkv = self.conv(p + ".qkv", x, act=False)
        q, k, v = qkv.view(B, heads, kd * 2 + hd, N).split([kd, kd, hd], dim=2)
        attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
        attn = attn.softmax(dim=-1)
        o = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + self.conv(p + ".pe", v.reshape(B, C, H, W), act=False)
        return self.conv(p + ".proj", o, act=False)

    def c2psa(self, x):
        a, b = self.conv("model.10.cv1", x).chunk(2, 1)
        for j in range(self.n2):
            m = "model.10.m.%d" % j
            b = b + self.attention(m + ".attn", b)
            b = b + self.conv(m + ".ffn.1", self.conv(m + ".ffn.0", b), act=False)
        return self.conv("model.10.cv2", torch.cat((a, b), 1))

    def forward(self, x):
        """x: [B,3,H,W] fp32 -> raw head output [B, 64+nc, A] (box logits then class logits)."""
        x0 = self.conv("model.0", x, 2)
        x1 = self.conv("model.1", x0, 2)
        x2 = self.c3k2(2, x1, False)
        x3 = self.conv("model.3", x2, 2)
        x4 = self.c3k2(4, x3, False)
        x5 = self.conv("model.5", x4, 2)
        x6 = self.c3k2(6, x5, True)
        x7 = self.conv("model.7", x6, 2)
        x8 = self.c3k2(8, x7, True)
        x9 = self.sppf(x8)
        x10 = self.c2psa(x9)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        x13 = self.c3k2(13, torch.cat((up(x10), x6), 1), False)
        x16 = self.c3k2(16, torch.cat((up(x13), x4), 1), False)
        x17 = self.conv("model.17", x16, 2)
        x19 = self.c3k2(19, torch.cat((x17, x13), 1), False)
        x20 = self.conv("model.20", x19, 2)
        x22 = self.c3k2(22, torch.cat((x20, x10), 1), True)
        outs = []
        for lvl, f in enumerate((x16, x19, x22)):
            b = "model.23.cv2.%d" % lvl
            bx = self.conv(b + ".2", self.conv(b + ".1", self.conv(b + ".0", f)), act=False)
            k = "model.23.cv3.%d" % lvl
            t = self.conv(k + ".0.1", self.conv(k + ".0.0", f))
            t = self.conv(k + ".1.1", self.conv(k + ".1.0", t))
            cl = self.conv(k + ".2", t, act=False)
            outs.append(torch.cat((bx, cl), 1))
        self.level_shapes = [tuple(o.shape[2:]) for o in outs]
        return torch.cat([o.flatten(2) for o in outs], 2)
