"""YOLO11 forward parity on the GPU: the plan of caesar_yolo_amd/yolo11_graph.py, shipped inside a CYW2
weight file and executed by the HIP runtime (incl. depth-wise conv and the C2PSA attention kernel), against the
independent torch restatement oracle/yolo11_ref.py.  Both are this repo's reading of the public ultralytics modules: no
ultralytics here, so parity with the real package is unpinned (see oracle/yolo11_ref.py)."""
import os
import numpy as np
import pytest
import torch
from gpu_common import netin_from_chw
from yolo11_common import seeded_folded

pytestmark = pytest.mark.gpu
# (convs whose output buffer is reused later, or that fuse a residual add, are not tapped: their buffer holds something else)
TAPS = ["model.1", "model.2.cv2", "model.9.cv2", "model.10.cv2", "model.13.cv2", "model.16.cv2", "model.19.cv2", "model.22.cv2",
        "model.23.cv3.0.0.0", "model.23.cv3.1.1.1"]


@pytest.mark.parametrize("scale,prec,tol_raw,tol_tap", [("n", "fp32", 2e-4, 1e-4), ("n", "fp16", 6e-2, 3e-2),
                                                        ("l", "fp32", 2e-4, 1e-4), ("l", "fp16", 6e-2, 3e-2)])
def test_yolo11_forward_matches_oracle(tmp_path, scale, prec, tol_raw, tol_tap):
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd.model import HipDetector
    from oracle import yolo11_ref as O
    nc = 3
    g, wd = seeded_folded(scale, nc)
    path = str(tmp_path / ("y11%s.cyw" % scale))
    W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], {0: "a", 1: "b", 2: "c"})
    assert W.read_cyw_header(path) == (scale, {0: "a", 1: "b", 2: "c"}, nc, len(g.convs))
    det = HipDetector(path, device=0, precision=prec, max_batch=2, max_imgsz=256)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.uniform(0, 1, (2, 3, 256, 192)).astype(np.float32))
    if prec == "fp16":
        wd = {k: (v[0].astype(np.float16).astype(np.float32), v[1]) for k, v in wd.items()}
        x = x.half().float()
    net = O.Net11(wd, scale, nc)
    net.taps = {}
    with torch.no_grad():
        raw = net.forward(x).permute(0, 2, 1).contiguous()          # [B, A, 64+nc]
    pred = det.forward(netin_from_chw(x, det.dtype))
    torch.cuda.synchronize()
    for name in TAPS:
        if name not in net.taps:
            continue
        ref = net.taps[name]
        got = torch.from_numpy(det.read_conv(name, ref.numel()))
        assert tuple(got.shape) == tuple(ref.shape), name
        sc = max(float(ref.abs().max()), 1.0)
        err = float((got - ref).abs().max())
        assert err <= tol_tap * sc, "%s: max abs err %.3e (scale %.2f)" % (name, err, sc)
    err = float((pred.cpu() - raw).abs().max())
    assert err <= tol_raw * max(1.0, float(raw.abs().max())), "raw head output: max abs err %.3e" % err


def test_yolo11_model_call_end_to_end(tmp_path):
    """The reference-shaped model call (`YOLO(weights)(image, imgsz=, conf=, iou=)`, caesar_yolo/evaluation.py:181-193) with a
    YOLO11 weight file: LetterBox -> network -> decode -> NMS -> scale_boxes on the GPU (f32 context) against the oracle
    (oracle/yolo11_ref.Net11 + the shared decode/NMS of oracle/yolov8_ref.py)."""
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd.model import YOLO
    from oracle import yolo11_ref as O
    from oracle import yolov8_ref as Y
    nc, names = 3, {0: "a", 1: "b", 2: "c"}
    g, wd = seeded_folded("n", nc, cls_bias=-1.5)
    path = str(tmp_path / "y11n.cyw")
    W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], names)
    model = YOLO(path, precision="fp32", max_batch=1, max_imgsz=256, device=0)
    assert model.names == names and model.scale == "n"
    oracle = Y.OracleYOLO(None, names, net=O.Net11(wd, "n", nc))
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, (200, 230, 3))
    got = model(img, imgsz=256, conf=0.3, iou=0.5)[0]
    ref = oracle(img, imgsz=256, conf=0.3, iou=0.5)[0]
    n = len(ref.boxes.conf)
    assert 3 <= n <= 300, n
    assert len(got.boxes.conf) == n
    np.testing.assert_array_equal(got.boxes.cls.cpu().numpy().astype(int), ref.boxes.cls.numpy().astype(int))
    np.testing.assert_allclose(got.boxes.conf.cpu().numpy(), ref.boxes.conf.numpy(), atol=1e-4)
    np.testing.assert_allclose(got.boxes.xyxy.cpu().numpy(), ref.boxes.xyxy.numpy(), atol=256 * 1e-4)
