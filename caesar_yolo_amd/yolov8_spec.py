"""YOLOv8 detection graph specification (host side): the ordered list of convolutions, their
tensor names and shapes for a (scale, nc) pair.

The network is what the reference instantiates with `YOLO(weights)` (scripts/run.py:347) and calls
at caesar_yolo/evaluation.py:181-193.  Its definition lives in the third-party `ultralytics`
package (unpinned in requirements.txt:9 and absent from the container), so this follows the public
yolov8.yaml graph as written down in SURVEY.md Appendix A.1/B.  Tensor names match the ultralytics
state_dict naming (`model.<i>...conv.weight`, `...bn.*`) so that real checkpoints can be mapped 1:1.

The C++ runtime (csrc/cy_plan.cpp) builds the same list independently; tests compare the two through
the C-ABI (`cy_plan_num_convs`, `cy_plan_conv_desc`).
"""
import math

SCALES = {  # depth, width, max_channels  (yolov8.yaml)
    "n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512),
}
REG_MAX = 16
DEFAULT_NAMES = {0: "spurious", 1: "compact", 2: "extended", 3: "extended-multisland", 4: "flagged"}


def _make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


def channels(scale):
    d, w, mc = SCALES[scale]
    ch = lambda c: _make_divisible(min(c, mc) * w, 8)
    rep = lambda n: max(round(n * d), 1)
    return {"c1": ch(64), "c2": ch(128), "c3": ch(256), "c4": ch(512), "c5": ch(1024),
            "n3": rep(3), "n6": rep(6)}


class ConvSpec(object):
    __slots__ = ("name", "cin", "cout", "k", "s", "act", "bn")

    def __init__(self, name, cin, cout, k, s, act=True, bn=True):
        self.name, self.cin, self.cout, self.k, self.s, self.act, self.bn = name, cin, cout, k, s, act, bn

    def as_tuple(self):
        return (self.name, self.cin, self.cout, self.k, self.s, int(self.act), int(self.bn))


def conv_list(scale="l", nc=5):
    """Canonical (state_dict) order: per module cv1, cv2, m.*; Detect: cv2.{0,1,2}.{0,1,2} then cv3."""
    c = channels(scale)
    c1, c2, c3, c4, c5, n3, n6 = c["c1"], c["c2"], c["c3"], c["c4"], c["c5"], c["n3"], c["n6"]
    out = []

    def conv(i, cin, cout, k, s):
        out.append(ConvSpec("model.%d" % i, cin, cout, k, s))

    def c2f(i, cin, cout, n):
        h = cout // 2
        out.append(ConvSpec("model.%d.cv1" % i, cin, 2 * h, 1, 1))
        out.append(ConvSpec("model.%d.cv2" % i, (2 + n) * h, cout, 1, 1))
        for j in range(n):
            out.append(ConvSpec("model.%d.m.%d.cv1" % (i, j), h, h, 3, 1))
            out.append(ConvSpec("model.%d.m.%d.cv2" % (i, j), h, h, 3, 1))

    conv(0, 3, c1, 3, 2)
    conv(1, c1, c2, 3, 2)
    c2f(2, c2, c2, n3)
    conv(3, c2, c3, 3, 2)
    c2f(4, c3, c3, n6)
    conv(5, c3, c4, 3, 2)
    c2f(6, c4, c4, n6)
    conv(7, c4, c5, 3, 2)
    c2f(8, c5, c5, n3)
    out.append(ConvSpec("model.9.cv1", c5, c5 // 2, 1, 1))
    out.append(ConvSpec("model.9.cv2", c5 * 2, c5, 1, 1))
    c2f(12, c5 + c4, c4, n3)
    c2f(15, c4 + c3, c3, n3)
    conv(16, c3, c3, 3, 2)
    c2f(18, c3 + c4, c4, n3)
    conv(19, c4, c4, 3, 2)
    c2f(21, c4 + c5, c5, n3)
    ch = (c3, c4, c5)
    cb = max(16, ch[0] // 4, REG_MAX * 4)
    cc = max(ch[0], min(nc, 100))
    for lvl in range(3):
        out.append(ConvSpec("model.22.cv2.%d.0" % lvl, ch[lvl], cb, 3, 1))
        out.append(ConvSpec("model.22.cv2.%d.1" % lvl, cb, cb, 3, 1))
        out.append(ConvSpec("model.22.cv2.%d.2" % lvl, cb, 4 * REG_MAX, 1, 1, act=False, bn=False))
    for lvl in range(3):
        out.append(ConvSpec("model.22.cv3.%d.0" % lvl, ch[lvl], cc, 3, 1))
        out.append(ConvSpec("model.22.cv3.%d.1" % lvl, cc, cc, 3, 1))
        out.append(ConvSpec("model.22.cv3.%d.2" % lvl, cc, nc, 1, 1, act=False, bn=False))
    return out


def conv_out_hw(h, w, k, s):
    p = k // 2
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def conv_macs(scale, nc, H, W):
    """Algorithmic multiply-accumulates of every conv for an H x W input."""
    c = channels(scale)
    res = {}
    h2, w2 = conv_out_hw(H, W, 3, 2)
    h4, w4 = conv_out_hw(h2, w2, 3, 2)
    h8, w8 = conv_out_hw(h4, w4, 3, 2)
    h16, w16 = conv_out_hw(h8, w8, 3, 2)
    h32, w32 = conv_out_hw(h16, w16, 3, 2)
    lvl_of = {0: (h2, w2), 1: (h4, w4), 2: (h4, w4), 3: (h8, w8), 4: (h8, w8), 5: (h16, w16),
              6: (h16, w16), 7: (h32, w32), 8: (h32, w32), 9: (h32, w32), 12: (h16, w16),
              15: (h8, w8), 16: (h16, w16), 18: (h16, w16), 19: (h32, w32), 21: (h32, w32)}
    det = {0: (h8, w8), 1: (h16, w16), 2: (h32, w32)}
    total = 0
    for cs in conv_list(scale, nc):
        parts = cs.name.split(".")
        idx = int(parts[1])
        if idx == 22:
            ho, wo = det[int(parts[3])]
        else:
            ho, wo = lvl_of[idx]
        m = ho * wo * cs.cout * cs.cin * cs.k * cs.k
        res[cs.name] = m
        total += m
    return total, res


def num_anchors(H, W):
    n = 0
    for s in (8, 16, 32):
        n += ((H + s - 1) // s) * ((W + s - 1) // s)
    return n
