"""Self-checks of the network oracle (oracle/yolov8_ref.py), which is PARITY UNPINNED with respect to ultralytics itself
(absent from the reference tree and from the container):

  * the graph built a SECOND, independent way -- torch.nn modules as ultralytics defines them (`Conv` = Conv2d(bias=False) +
    BatchNorm2d(eps=1e-3) + SiLU, `Bottleneck`, `C2f`, `SPPF`, `Detect` with its DFL branch layout; the public yolov8.yaml,
    SURVEY.md Appendix A.1 step 4), UN-fused, in eval mode -- gives the same raw head output as `oracle.yolov8_ref.Net`
    run on the folded (w * gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var + eps)) weights: pins the wiring (chunk /
    concat order, shortcut flags, upsample + concat, head branches) and the BN folding formula against each other;
  * the published yolov8l size (43.7 M parameters / 165.2 GFLOPs at 640x640, nc = 80) computed from the ORACLE's own layers;
  * letterbox geometry on the shapes SURVEY.md Appendix A.1 step 2 lists.
"""
import math
import numpy as np
import pytest
import torch
import torch.nn as nn
from oracle import yolov8_ref as Y


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c2, 3, 1), Conv(c2, c2, 3, 1)          # C2f uses e = 1.0, k = (3, 3)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in self.m)
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c1 // 2, 1, 1), Conv(c1 // 2 * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        y = [self.cv1(x)]
        y.extend(self.m(y[-1]) for _ in range(3))
        return self.cv2(torch.cat(y, 1))


class Detect(nn.Module):
    def __init__(self, nc, ch):
        super().__init__()
        c2, c3 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 64, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)

    def forward(self, xs):
        return [torch.cat((self.cv2[i](x), self.cv3[i](x)), 1) for i, x in enumerate(xs)]


class V8(nn.Module):
    """yolov8.yaml as a module list: [from, repeats, module, args]"""

    def __init__(self, scale, nc):
        super().__init__()
        d, w, mc = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.5, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512), "x": (1.0, 1.25, 512)}[scale]
        ch = lambda c: make_divisible(min(c, mc) * w, 8)
        rep = lambda n: max(round(n * d), 1)
        L = [Conv(3, ch(64), 3, 2), Conv(ch(64), ch(128), 3, 2), C2f(ch(128), ch(128), rep(3), True),
             Conv(ch(128), ch(256), 3, 2), C2f(ch(256), ch(256), rep(6), True),
             Conv(ch(256), ch(512), 3, 2), C2f(ch(512), ch(512), rep(6), True),
             Conv(ch(512), ch(1024), 3, 2), C2f(ch(1024), ch(1024), rep(3), True), SPPF(ch(1024), ch(1024), 5),
             nn.Upsample(None, 2, "nearest"), None, C2f(ch(1024) + ch(512), ch(512), rep(3)),
             nn.Upsample(None, 2, "nearest"), None, C2f(ch(512) + ch(256), ch(256), rep(3)),
             Conv(ch(256), ch(256), 3, 2), None, C2f(ch(256) + ch(512), ch(512), rep(3)),
             Conv(ch(512), ch(512), 3, 2), None, C2f(ch(512) + ch(1024), ch(1024), rep(3)),
             Detect(nc, (ch(256), ch(512), ch(1024)))]
        self.model = nn.ModuleList(m if m is not None else nn.Identity() for m in L)
        self.cat = {11: (10, 6), 14: (13, 4), 17: (16, 12), 20: (19, 9)}

    def forward(self, x):
        y = []
        for i, m in enumerate(self.model):
            if i in self.cat:
                x = torch.cat([y[j] for j in self.cat[i]], 1)
            elif i == 22:
                x = m([y[15], y[18], y[21]])
            else:
                x = m(x)
            y.append(x)
        return torch.cat([o.flatten(2) for o in x], 2)


def folded_weights(net):
    """state_dict of the un-fused modules -> the oracle's {conv name: (W, b)} with BatchNorm folded (A.1 step 4)."""
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    out = {}
    for k in sd:
        if k.endswith(".conv.weight"):
            base = k[:-len(".conv.weight")]
            g, b, mu, var = (sd[base + ".bn." + n] for n in ("weight", "bias", "running_mean", "running_var"))
            sc = g / torch.sqrt(var + 1e-3)
            out[base] = ((sd[k] * sc.view(-1, 1, 1, 1)).float().numpy(), (b - mu * sc).float().numpy())
        elif k.endswith(".weight") and sd[k].ndim == 4 and (k[:-len(".weight")] + ".bias") in sd:     # Detect's plain Conv2d
            base = k[:-len(".weight")]
            out[base] = (sd[k].float().numpy(), sd[base + ".bias"].float().numpy())
    return out


@pytest.mark.parametrize("scale,nc,hw", [("n", 5, (96, 64)), ("s", 3, (64, 64))])
def test_unfused_module_graph_equals_oracle_on_folded_weights(scale, nc, hw):
    torch.manual_seed(7)
    net = V8(scale, nc)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):                       # non-trivial statistics: the folding must matter
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
    net.eval()
    wd = folded_weights(net)
    oracle = Y.Net(wd, scale, nc)
    assert set(wd) == set(oracle.w) and len(wd) == 63          # 63 convolutions at depth 0.33
    x = torch.rand(2, 3, *hw)
    with torch.no_grad():
        ref = net(x)
        got = oracle.forward(x)
    assert got.shape == ref.shape == (2, 64 + nc, sum((hw[0] // s) * (hw[1] // s) for s in (8, 16, 32)))
    err = float((got - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err


def test_published_yolov8l_size_from_the_oracle_layers():
    """ultralytics publishes 43.7 M parameters / 165.2 GFLOPs for yolov8l (nc = 80, 640x640): count them on the oracle's
    own forward pass (shapes from caesar_yolo_amd.yolov8_spec are only used to allocate zero weights)."""
    from caesar_yolo_amd import yolov8_spec as S
    wd = {c.name: (np.zeros((c.cout, c.cin, c.k, c.k), np.float32), np.zeros(c.cout, np.float32)) for c in S.conv_list("l", 80)}
    net = Y.Net(wd, "l", 80)
    macs = [0]
    conv0 = net.conv

    def counting(name, x, s=1, act=True):
        y = conv0(name, x, s, act)
        w = net.w[name][0]
        macs[0] += y.shape[2] * y.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        return y
    net.conv = counting
    with torch.no_grad():
        out = net.forward(torch.zeros(1, 3, 640, 640))
    assert out.shape == (1, 64 + 80, 8400)
    params = sum(w.numel() + b.numel() for w, b in net.w.values())
    assert abs(params / 1e6 - 43.67) < 0.05
    assert abs(2 * macs[0] / 1e9 - 165.1) < 0.3


@pytest.mark.parametrize("h0,w0,imgsz,want", [(132, 132, 640, (640, 640, 0, 0, 640, 640)), (512, 512, 512, (512, 512, 0, 0, 512, 512)),
                                               (512, 394, 512, (512, 394, 0, 11, 512, 416)), (394, 512, 512, (394, 512, 11, 0, 416, 512)),
                                               (394, 394, 512, (512, 512, 0, 0, 512, 512))])
def test_letterbox_geometry(h0, w0, imgsz, want):
    nh, nw, top, bottom, left, right, H, W = Y.letterbox_params(h0, w0, imgsz)
    assert (nh, nw, top, left, H, W) == want
