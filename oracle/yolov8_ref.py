"""CPU oracle for the detector call `model(image, imgsz=, conf=, iou=)` -- torch-CPU fp32.
TEST INFRASTRUCTURE ONLY (importers: tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

PARITY UNPINNED: the reference delegates this call (caesar_yolo/evaluation.py:181-193, results read at
:261-265) to the third-party `ultralytics` package, which is unpinned (requirements.txt:9, setup.py:34),
absent from /root/reference and from the container, as are cv2/torchvision and any trained weights.
Nothing in the reference's tests pins letterbox, forward or NMS outputs.  This file restates the public
ultralytics-8.x detect pipeline as written down in SURVEY.md Appendix A.1 (LetterBox, /255, the
yolov8.yaml graph with Conv+BN folded, Detect/DFL decode, non_max_suppression with torchvision-style
nms, scale_boxes).  Self-consistency pins (tests/test_oracle_net.py): the graph rebuilt independently as
un-fused torch.nn modules (Conv2d + BatchNorm2d(eps=1e-3) + SiLU, C2f, SPPF, Detect) gives the same output on
the folded weights; parameter/FLOP counts of THIS graph reproduce the published yolov8l figures; letterbox
geometry on the shapes SURVEY.md Appendix A.1 lists.

Everything is fp32 like the reference `--devices=cpu` run; the input image is float64 HWC in [0,255]
exactly as Analyzer hands it over.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

REG_MAX = 16
MAX_WH = 7680.0
MAX_NMS = 30000
MAX_DET = 300


# --------------------------------------------------------------------------- letterbox
def letterbox_params(h0, w0, imgsz, stride=32):
    """ultralytics LetterBox(new_shape=imgsz, auto=True, scaleup=True, center=True) geometry.
    Returns (new_h, new_w, top, bottom, left, right, H, W)."""
    r = min(imgsz / h0, imgsz / w0)
    new_w, new_h = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_h, new_w, top, bottom, left, right, new_h + top + bottom, new_w + left + right


def resize_linear(img, new_h, new_w):
    """cv2.resize(..., INTER_LINEAR) restated for float64 HWC input: half-pixel centres, source index and
    weight computed in float32 as cv2 does, edges clamped, arithmetic in double (cv2 absent: unpinned)."""
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img

    def axis(n_src, n_dst):
        scale = n_src / n_dst
        f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0[lo] = 0
        a[lo] = 0.0
        hi = i0 >= n_src - 1
        i0[hi] = n_src - 1
        a[hi] = 0.0
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, a.astype(np.float64)
    y0, y1, ay = axis(h, new_h)
    x0, x1, ax = axis(w, new_w)
    rows = img[:, x0] * (1.0 - ax)[None, :, None] + img[:, x1] * ax[None, :, None]
    return rows[y0] * (1.0 - ay)[:, None, None] + rows[y1] * ay[:, None, None]


def preprocess(image_hwc, imgsz):
    """LetterBox -> [..., ::-1] -> CHW -> float32 -> /255  (Appendix A.1 steps 2-3)."""
    h0, w0 = image_hwc.shape[:2]
    nh, nw, top, bottom, left, right, H, W = letterbox_params(h0, w0, imgsz)
    im = resize_linear(np.asarray(image_hwc, np.float64), nh, nw)
    out = np.full((H, W, 3), 114.0)
    out[top:top + nh, left:left + nw] = im
    out = out[..., ::-1].transpose(2, 0, 1)
    t = torch.from_numpy(np.ascontiguousarray(out)).float() / 255
    return t.unsqueeze(0), (H, W)


# --------------------------------------------------------------------------- network
class Net(object):
    """yolov8 detection graph on folded (conv weight, bias) pairs.  `weights`: name -> (W[co,ci,k,k], b[co])."""

    def __init__(self, weights, scale="l", nc=5):
        self.w = {k: (torch.as_tensor(v[0], dtype=torch.float32), torch.as_tensor(v[1], dtype=torch.float32))
                  for k, v in weights.items()}
        self.nc = nc
        self.n3 = max(round(3 * {"n": .33, "s": .33, "m": .67, "l": 1., "x": 1.}[scale]), 1)
        self.n6 = max(round(6 * {"n": .33, "s": .33, "m": .67, "l": 1., "x": 1.}[scale]), 1)
        self.taps = None          # optional dict collecting intermediate activations

    def conv(self, name, x, s=1, act=True):
        w, b = self.w[name]
        y = F.conv2d(x, w, b, stride=s, padding=w.shape[-1] // 2)
        y = F.silu(y) if act else y
        if self.taps is not None:
            self.taps[name] = y
        return y

    def c2f(self, i, x, n, shortcut):
        y = list(self.conv("model.%d.cv1" % i, x).chunk(2, 1))
        for j in range(n):
            t = self.conv("model.%d.m.%d.cv2" % (i, j), self.conv("model.%d.m.%d.cv1" % (i, j), y[-1]))
            y.append(y[-1] + t if shortcut else t)
        return self.conv("model.%d.cv2" % i, torch.cat(y, 1))

    def sppf(self, x):
        a = self.conv("model.9.cv1", x)
        b = F.max_pool2d(a, 5, 1, 2)
        c = F.max_pool2d(b, 5, 1, 2)
        d = F.max_pool2d(c, 5, 1, 2)
        return self.conv("model.9.cv2", torch.cat((a, b, c, d), 1))

    def forward(self, x):
        """x: [B,3,H,W] fp32 -> raw head output [B, 64+nc, A] (box logits then class logits)."""
        x0 = self.conv("model.0", x, 2)
        x1 = self.conv("model.1", x0, 2)
        x2 = self.c2f(2, x1, self.n3, True)
        x3 = self.conv("model.3", x2, 2)
        x4 = self.c2f(4, x3, self.n6, True)
        x5 = self.conv("model.5", x4, 2)
        x6 = self.c2f(6, x5, self.n6, True)
        x7 = self.conv("model.7", x6, 2)
        x8 = self.c2f(8, x7, self.n3, True)
        x9 = self.sppf(x8)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        x12 = self.c2f(12, torch.cat((up(x9), x6), 1), self.n3, False)
        x15 = self.c2f(15, torch.cat((up(x12), x4), 1), self.n3, False)
        x16 = self.conv("model.16", x15, 2)
        x18 = self.c2f(18, torch.cat((x16, x12), 1), self.n3, False)
        x19 = self.conv("model.19", x18, 2)
        x21 = self.c2f(21, torch.cat((x19, x9), 1), self.n3, False)
        outs = []
        for lvl, f in enumerate((x15, x18, x21)):
            bx = self.conv("model.22.cv2.%d.2" % lvl, self.conv("model.22.cv2.%d.1" % lvl, self.conv(
                "model.22.cv2.%d.0" % lvl, f)), act=False)
            cl = self.conv("model.22.cv3.%d.2" % lvl, self.conv("model.22.cv3.%d.1" % lvl, self.conv(
                "model.22.cv3.%d.0" % lvl, f)), act=False)
            outs.append(torch.cat((bx, cl), 1))
        self.level_shapes = [tuple(o.shape[2:]) for o in outs]
        return torch.cat([o.flatten(2) for o in outs], 2)


def make_anchors(level_shapes, strides=(8, 16, 32)):
    pts, st = [], []
    for (h, w), s in zip(level_shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + 0.5
        sy = torch.arange(h, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts).t(), torch.cat(st).t()     # [2,A], [1,A]


def decode(raw, level_shapes, nc):
    """Detect inference branch: DFL expectation, dist2bbox(xywh) * stride, sigmoid(cls) -> [B, 4+nc, A]."""
    b = raw.shape[0]
    box, cls = raw.split((4 * REG_MAX, nc), 1)
    a = box.shape[-1]
    p = box.view(b, 4, REG_MAX, a).transpose(2, 1).softmax(1)          # [B,16,4,A]
    proj = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    dist = (p * proj).sum(1)                                            # [B,4,A]
    anc, strides = make_anchors(level_shapes)
    lt, rb = dist.chunk(2, 1)
    x1y1 = anc.unsqueeze(0) - lt
    x2y2 = anc.unsqueeze(0) + rb
    c_xy = (x1y1 + x2y2) / 2
    wh = x2y2 - x1y1
    dbox = torch.cat((c_xy, wh), 1) * strides
    return torch.cat((dbox, cls.sigmoid()), 1)


def nms_indices(boxes, scores, iou_thr):
    """torchvision.ops.nms semantics: stable sort by score descending, suppress when IoU > thr (strict)."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = b.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        dead[i + 1:] |= ovr > iou_thr
    return order[torch.as_tensor(keep, dtype=torch.long)]


def non_max_suppression(pred, conf_thres, iou_thres, nc):
    """ultralytics ops.non_max_suppression (agnostic=False, multi_label=False, max_det=300, max_nms=30000,
    max_wh=7680).  pred: [B, 4+nc, A].  Returns per image (det [n,6], candidate anchor index [n])."""
    xc = pred[:, 4:4 + nc].amax(1) > conf_thres
    pred = pred.transpose(-1, -2).clone()
    xy, wh = pred[..., :2].clone(), pred[..., 2:4].clone()
    pred[..., :2] = xy - wh / 2
    pred[..., 2:4] = xy + wh / 2
    out = []
    for xi in range(pred.shape[0]):
        aidx = torch.nonzero(xc[xi]).flatten()
        x = pred[xi][xc[xi]]
        box, cls = x[:, :4], x[:, 4:4 + nc]
        if x.shape[0] == 0:
            out.append((torch.zeros((0, 6)), aidx))
            continue
        conf, j = cls.max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)
        sel = conf.view(-1) > conf_thres
        x, aidx = x[sel], aidx[sel]
        if x.shape[0] > MAX_NMS:
            o = x[:, 4].argsort(descending=True, stable=True)[:MAX_NMS]
            x, aidx = x[o], aidx[o]
        c = x[:, 5:6] * MAX_WH
        i = nms_indices(x[:, :4] + c, x[:, 4], iou_thres)[:MAX_DET]
        out.append((x[i], aidx[i]))
    return out


def scale_boxes(boxes, img1_hw, img0_hw):
    """ultralytics ops.scale_boxes + clip_boxes (padding=True, xywh=False)."""
    gain = min(img1_hw[0] / img0_hw[0], img1_hw[1] / img0_hw[1])
    padw = round((img1_hw[1] - img0_hw[1] * gain) / 2 - 0.1)
    padh = round((img1_hw[0] - img0_hw[0] * gain) / 2 - 0.1)
    boxes = boxes.clone()
    boxes[:, 0] -= padw
    boxes[:, 1] -= padh
    boxes[:, 2] -= padw
    boxes[:, 3] -= padh
    boxes[:, :4] /= gain
    boxes[:, 0].clamp_(0, img0_hw[1])
    boxes[:, 1].clamp_(0, img0_hw[0])
    boxes[:, 2].clamp_(0, img0_hw[1])
    boxes[:, 3].clamp_(0, img0_hw[0])
    return boxes


class OracleYOLO(object):
    """Stand-in for `ultralytics.YOLO(weights)`: `.names` and `__call__(img, imgsz=, conf=, iou=, **ignored)`."""

    def __init__(self, weights, names, scale="l", threads=None, net=None):
        self.names = dict(names)
        self.net = net if net is not None else Net(weights, scale, len(names))      # net: e.g. yolo11_ref.Net11 (same head layout)
        if threads:
            torch.set_num_threads(threads)

    @torch.no_grad()
    def predict_raw(self, image_hwc, imgsz=640, conf=0.25, iou=0.7):
        x, hw = preprocess(image_hwc, imgsz)
        raw = self.net.forward(x)
        pred = decode(raw, self.net.level_shapes, self.net.nc)
        det, aidx = non_max_suppression(pred, conf, iou, self.net.nc)[0]
        det = det.clone()
        det[:, :4] = scale_boxes(det[:, :4], hw, image_hwc.shape[:2])
        return det, aidx, raw, pred

    def __call__(self, image_hwc, imgsz=640, conf=0.25, iou=0.7, **kw):
        det, _, _, _ = self.predict_raw(image_hwc, imgsz, conf, iou)

        class _R(object):
            pass
        r = _R()
        r.boxes = _R()
        r.boxes.xyxy, r.boxes.conf, r.boxes.cls = det[:, :4], det[:, 4], det[:, 5]
        return [r]
