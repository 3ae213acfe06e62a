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
