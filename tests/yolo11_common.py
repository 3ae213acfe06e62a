"""Shared helpers of the YOLO11 tests: seeded random weights for a yolo11_graph.Graph, calibrated layer by layer with the
CPU oracle so that activations stay O(1) through ~90-170 convolutions (test infrastructure; uses oracle/)."""
import numpy as np
import torch
import torch.nn.functional as F
from caesar_yolo_amd import yolo11_graph as G
from oracle import yolo11_ref as O


class _Calibrating(O.Net11):
    """One sequential pass: every conv's weight is rescaled so that its pre-activation has unit std on the probe input."""
    def conv(self, name, x, s=1, act=True):
        w, b = self.w[name]
        groups = x.shape[1] // w.shape[1]
        y0 = F.conv2d(x, w, None, stride=s, padding=w.shape[-1] // 2, groups=groups)
        sd = float(y0.std())
        w = w / (sd if sd > 0 else 1.0)
        self.w[name] = (w, b)
        y = F.conv2d(x, w, b, stride=s, padding=w.shape[-1] // 2, groups=groups)
        return F.silu(y) if act else y


def seeded_folded(scale, nc, seed=11, probe_hw=128, cls_bias=None):
    """-> (graph, {conv name: (W, b)} folded fp32 numpy).  cls_bias: constant bias of the class-logit convs (negative ->
    few candidates above the confidence threshold, like a trained detector)."""
    g = G.build(scale, nc)
    rng = np.random.default_rng(seed)
    wd = {}
    for cs in g.convs:
        fan = (cs.cin // cs.groups) * cs.k * cs.k
        w = (rng.standard_normal((cs.cout, cs.cin // cs.groups, cs.k, cs.k)) / np.sqrt(fan)).astype(np.float32)
        b = (rng.standard_normal(cs.cout) * 0.1).astype(np.float32)
        wd[cs.name] = (w, b)
    net = _Calibrating(wd, scale, nc)
    x = torch.from_numpy(rng.uniform(0, 1, (1, 3, probe_hw, probe_hw)).astype(np.float32))
    with torch.no_grad():
        net.forward(x)
    out = {k: (v[0].numpy().copy(), v[1].numpy().copy()) for k, v in net.w.items()}
    if cls_bias is not None:
        for l in range(3):
            k = "model.23.cv3.%d.2" % l
            out[k] = (out[k][0], np.full_like(out[k][1], cls_bias))
    return g, out
