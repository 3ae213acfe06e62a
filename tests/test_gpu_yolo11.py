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


@pytest.mark.parametrize("scale,prec,tol_raw,tol_tap", [("n", "fp32", 2e-4, 1e-4), ("n", "fp16", 6e-2, 3e-2), ("n", "fp16x3", 2e-4, 1e-4),
                                                        ("l", "fp32", 2e-4, 1e-4), ("l", "fp16", 6e-2, 3e-2), ("l", "fp16x3", 2e-4, 1e-4)])
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


def test_seeded_yolo11_weights_of_the_benchmark():
    """`YOLO("seeded11:<scale>:<nc>")` (bench.py --weights seeded11:l:5): the product-side seeded YOLO11 weights -- the random draw
    normalised by the tabulated per-conv factors, fp16-valued like a checkpoint on disk -- load, run every MFMA layer of the fp16x3
    context on the two-pass form, keep activations O(1) and match the oracle (2e-4)."""
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd.model import YOLO
    from oracle import yolo11_ref as O
    g, wd = W.seeded11_folded("n", 5)
    model = YOLO("seeded11:n:5", precision="fp16x3", max_batch=2, max_imgsz=256, device=0)
    det = model.engine(0)
    n2, n3 = det.weight_passes()
    assert n3 == 0 and n2 > 60, (n2, n3)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.uniform(0, 1, (2, 3, 256, 224)).astype(np.float32))
    with torch.no_grad():
        raw = O.Net11(wd, "n", 5).forward(x).permute(0, 2, 1).contiguous()
    pred = det.forward(netin_from_chw(x, det.dtype)).cpu()
    sc = max(1.0, float(raw.abs().max()))
    err = float((pred - raw).abs().max())
    print("seeded11:n:5 fp16x3 raw head output vs oracle: %.3e (scale %.2f; %d two-pass layers)" % (err, sc, n2))
    assert 1.0 < sc < 100.0 and err <= 2e-4 * sc
    det.close()


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


def test_yolo11_tile_path(tmp_path):
    """The batched tile entry (crop -> device preprocessing -> YOLO11 -> NMS -> IoU merge) with a CYW2 file: per-tile results
    equal the CPU oracle's chain (numpy preprocessing, oracle/yolo11_ref network, shared NMS and process_detections)."""
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd import preprocessing as PP
    from caesar_yolo_amd.model import YOLO
    from gpu_common import ROOT
    from oracle import yolo11_ref as O
    from oracle import yolov8_ref as Y
    from oracle import preprocessing_ref as P
    from oracle import postproc_ref as R
    nc, names = 3, {0: "a", 1: "b", 2: "c"}
    conf, iou, soft, hard = 0.3, 0.5, 0.3, 0.8
    g, wd = seeded_folded("n", nc, cls_bias=-1.5)
    path = str(tmp_path / "y11n.cyw")
    W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], names)
    model = YOLO(path, precision="fp32", max_batch=4, max_imgsz=256, device=0)
    oracle = Y.OracleYOLO(None, names, net=O.Net11(wd, "n", nc))
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"].astype(np.float32)
    mosaic = model.engine().mosaic_to_device(img)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    coords = [(0, 256, 0, 256), (256, 512, 0, 256), (128, 384, 128, 384), (600, 856, 400, 656)]
    res = model.predict_tiles(mosaic, coords, cfg, imgsz=256, conf=conf, iou=iou,
                              merge_overlap_iou_thr_soft=soft, merge_overlap_iou_thr_hard=hard)
    total = 0
    for (x0, x1, y0, y1), r in zip(coords, res):
        im = dp(P.to_cube(np.array(img[y0:y1, x0:x1], np.float32)))
        det, _, _, _ = oracle.predict_raw(im, 256, conf, iou)
        kb, ks, kc = R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), conf, soft, hard)[:3]
        assert r is not None and len(r.boxes.conf) == len(ks)
        total += len(ks)
        if len(ks):
            np.testing.assert_allclose(r.boxes.conf.cpu().numpy(), ks, atol=1e-4)
            np.testing.assert_array_equal(r.boxes.cls.cpu().numpy().astype(int), np.asarray(kc).astype(int))
            np.testing.assert_allclose(r.boxes.xyxy.cpu().numpy(), kb, atol=256 * 1e-4)
    assert total >= 4


@pytest.mark.parametrize("field,delta", [("out_coff", 8), ("in0_coff", 8), ("res_coff", 1 << 20), ("c0", 64)])
def test_malformed_cyw2_plan_is_refused(tmp_path, field, delta):
    """The execution plan travels inside the weight file (CYW2): an op whose channel slice leaves its tensor must be refused
    at load time, not make a kernel address outside the workspace."""
    import copy
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd import lib as L
    from caesar_yolo_amd.model import HipDetector
    g, wd = seeded_folded("n", 3)
    g = copy.deepcopy(g)
    k = [i for i, o in enumerate(g.ops) if o["kind"] == 1 and o["out"] >= 0 and (field != "res_coff" or o["res"] >= 0)][3]
    g.ops[k][field] += delta
    path = str(tmp_path / "bad.cyw")
    W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], {0: "a", 1: "b", 2: "c"})
    with pytest.raises(L.CyError, match="malformed CYW2 plan"):
        HipDetector(path, device=0, precision="fp16", max_batch=1, max_imgsz=64)


@pytest.mark.parametrize("scale,prec", [("l", "fp16"), ("l", "fp16x3"), ("n", "fp16")])
def test_narrow_1x1_layers_on_the_streaming_kernel_are_bit_identical(tmp_path, scale, prec, monkeypatch):
    """1x1 convolutions with <= 64 output channels inside the network (YOLO11l: the C3k branches of model.2 / model.4, 64 -> 32 and
    64 -> 64 with SiLU, fp16 or high / low halves into channel slices) run on the weights-stationary streaming kernel of the detect
    head (head1x1_kernel) instead of the generic 128 px x 64 ch tile; CY_NARROW_DIRECT=0 keeps the generic kernel.  Same MFMA chain
    and epilogue per value: the head output is bit-identical."""
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd.model import HipDetector
    g, wd = seeded_folded(scale, 3)
    path = str(tmp_path / ("y11%s.cyw" % scale))
    W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], {0: "a", 1: "b", 2: "c"})
    det = HipDetector(path, device=0, precision=prec, max_batch=3, max_imgsz=256)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.uniform(0, 1, (3, 3, 256, 192)).astype(np.float32))
    xin = netin_from_chw(x, det.dtype)
    monkeypatch.setenv("CY_NARROW_DIRECT", "0")
    p0 = det.forward(xin).cpu().clone()
    monkeypatch.setenv("CY_NARROW_DIRECT", "1")
    p1 = det.forward(xin).cpu().clone()
    assert torch.isfinite(p1).all()
    assert torch.equal(p0, p1), "max difference %.3e" % float((p0 - p1).abs().max())
    det.close()
