"""Post-processing parity through the C-ABI.
 * cy_decode_nms on the ORACLE's raw head output: kept-anchor index list identical (order included), boxes/scores within
   1e-4 * max(1, |x|) -- isolates decode/NMS/scale_boxes from network rounding.
 * cy_iou_merge against the golden vectors captured from the reference's Analyzer.process_detections: bit-exact.
 * end to end in the f32 context (letterbox_pack -> forward -> decode_nms) against the oracle model call: same number of
   boxes, same classes in the same order, scores within 1e-4, boxes within 1e-4 in NORMALISED image coordinates
   (|dx| <= 1e-4 * max(H, W) of the network input, the unit of ultralytics' own xyxyn).  Two fp32 evaluations of a
   100-layer network that sum in different orders cannot agree to 1e-4 PIXELS on 600-px coordinates (that is 2.5 ulp);
   the error actually observed is printed and recorded in DESIGN.md (about 2e-3 px).  Skipped, with the count reported,
   if a candidate sits within 1e-5 of the confidence threshold (discontinuities are reported, not hidden)."""
import os
import numpy as np
import pytest
import torch
from gpu_common import detector, oracle_model, netin_from_chw, assert_same_detections, ROOT

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _prep(name, h=None, w=None):
    from oracle import preprocessing_ref as P
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/" + name]
    if h:
        img = img[:h, :w]
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    return dp(P.to_cube(img))


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= TOL * np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("name,imgsz,conf,iou", [("big512", 512, 0.7, 0.5), ("syn192", 192, 0.5, 0.5),
                                                 ("galaxy", 640, 0.7, 0.5), ("rag", 256, 0.25, 0.7),
                                                 # conf ~ 0 at imgsz 640: nearly all 8400 anchors are candidates (> 8192: the
                                                 # global-memory sort of nms_kernel; ultralytics' max_nms = 30000 does not bind)
                                                 ("galaxy", 640, 0.001, 0.7)])
def test_decode_nms_on_oracle_logits(name, imgsz, conf, iou):
    det = detector("fp32", max_imgsz=640)
    m = oracle_model()
    img = _prep(name)
    d_ref, a_ref, raw, _ = m.predict_raw(img, imgsz, conf, iou)
    H, W = [s * 8 for s in m.net.level_shapes[0]]
    pred = raw.permute(0, 2, 1).contiguous().cuda()
    d, anch, cnt = det.decode_nms(pred, H, W, img.shape[0], img.shape[1], conf, iou)
    torch.cuda.synchronize()
    n = int(cnt[0])
    assert n == d_ref.shape[0]
    assert anch[0, :n].cpu().tolist() == a_ref.tolist()          # kept-box index set AND order
    assert _close(d[0, :n, :4].cpu().numpy(), d_ref[:, :4].numpy())
    assert _close(d[0, :n, 4].cpu().numpy(), d_ref[:, 4].numpy())
    assert np.array_equal(d[0, :n, 5].cpu().numpy(), d_ref[:, 5].numpy())


def test_iou_merge_matches_reference_golden():
    det = detector("fp32")
    g = np.load(os.path.join(ROOT, "tests/golden/process_detections.npz"))
    keys = sorted({k.rsplit("/", 1)[0] for k in g.files})
    assert len(keys) == 32
    for k in keys:
        soft, hard, sthr = g[k + "/thr"]
        b, s, c = g[k + "/in_xyxy"], g[k + "/in_conf"], g[k + "/in_cls"]
        n = len(s)
        d = torch.zeros((1, 300, 6), dtype=torch.float32)
        d[0, :n, :4] = torch.from_numpy(b.reshape(-1, 4))
        d[0, :n, 4] = torch.from_numpy(s)
        d[0, :n, 5] = torch.from_numpy(c)
        out, ocnt, osrc = det.iou_merge(d.cuda(), torch.tensor([n], dtype=torch.int32).cuda(), float(np.float32(sthr)),
                                        float(soft), float(hard))
        torch.cuda.synchronize()
        m = int(ocnt[0])
        assert m == len(g[k + "/out_conf"]), k
        o = out[0, :m].cpu().numpy()
        assert np.array_equal(o[:, :4], g[k + "/out_xyxy"]), k
        assert np.array_equal(o[:, 4], g[k + "/out_conf"]), k
        assert np.array_equal(o[:, 5].astype(np.int32), g[k + "/out_cls"]), k


# measured on MI355X (printed by the test): fp32 context <= 2.1e-3 px / 7e-6 score.  The asserted bounds are those figures with a
# factor ~2.5 of head-room, NOT the 1e-4 * max(H, W) (0.064 px) the normalised north-star tolerance would allow.
E2E_BOX_PX, E2E_SCORE = 5e-3, 2e-5


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
@pytest.mark.parametrize("name,imgsz,conf", [("big512", 512, 0.7), ("galaxy", 640, 0.7), ("syn192", 192, 0.7),
                                             # the smallest and largest sizes the reference publishes weights for (README.md:190-207):
                                             # the 512-px frame resized down by 4 / up by 2 in the letterbox (thresholds chosen on the
                                             # oracle: 150 boxes at 128 px, where the seeded weights score <= 0.046; 300 at 1024 px
                                             # with no candidate within 1e-5 of the threshold)
                                             ("big512", 128, 0.04), ("big512", 1024, 0.65)])
def test_model_call_end_to_end_fp32(name, imgsz, conf, prec):
    """The `model(image, imgsz=, conf=, iou=)` surface of caesar_yolo/evaluation.py:181-193 on the two parity contexts (exact
    fp32, and fp16x3 = fp16 high + low halves on the tuned kernels): same boxes in the same order, classes equal, boxes within
    5e-3 px, scores within 2e-5."""
    from caesar_yolo_amd.model import YOLO
    from gpu_common import seeded_weights
    iou = 0.5
    img = _prep(name)
    m = oracle_model()
    # a candidate within 1e-5 of the confidence threshold may legitimately flip (the score is the last bit of a 100-layer sum): the
    # threshold of the case is moved by a few 1e-3 until the oracle has no such candidate, so that every case compares a full list
    for dconf in (0.0, 0.003, -0.003, 0.006, -0.006):
        d_ref, a_ref, raw, pred_ref = m.predict_raw(img, imgsz, conf + dconf, iou)
        if int(np.sum(np.abs(pred_ref[0, 4:].amax(0).numpy() - (conf + dconf)) < 1e-5)) == 0:
            conf = conf + dconf
            break
    y = YOLO(seeded_weights()[0], precision=prec, max_batch=2, max_imgsz=max(640, imgsz), device=0)
    assert y.names == m.names
    r = y(img, device="cuda:0", imgsz=imgsz, conf=conf, iou=iou, save=False, visualize=False, show=False)[0]
    xyxy, cf, cl = r.boxes.xyxy.cpu().numpy(), r.boxes.conf.cpu().numpy(), r.boxes.cls.cpu().numpy()
    # discontinuity report: candidates within 1e-5 of the threshold can legitimately flip
    sc = pred_ref[0, 4:].amax(0).numpy()
    near = int(np.sum(np.abs(sc - conf) < 1e-5))
    if near == 0:
        assert len(cf) == d_ref.shape[0]
        H, W = [s * 8 for s in m.net.level_shapes[0]]
        # (coordinates of a 1024-px network input carry 1.6x the fp32 ulp, and the letterbox resizes the frame up by 2: 1.2e-2 px =
        # 1.2e-5 of the image size there; measured 1.07e-2 on the fp16-valued seeded weights of round 4)
        btol = E2E_BOX_PX * max(1.0, imgsz / 640.0) * (1.5 if imgsz > 640 else 1.0)
        moved = assert_same_detections(xyxy, cf, cl, d_ref[:, :4].numpy(), d_ref[:, 4].numpy(), d_ref[:, 5].numpy(), btol, E2E_SCORE,
                                       "%s %s@%d" % (prec, name, imgsz))
        if moved == 0:
            berr = float(np.abs(xyxy - d_ref[:, :4].numpy()).max()) if len(cf) else 0.0
            serr = float(np.abs(cf - d_ref[:, 4].numpy()).max()) if len(cf) else 0.0
            print("end-to-end %s %s@%d: %d boxes, max |dbox| = %.3e px (%.2e normalised), max |dscore| = %.3e"
                  % (prec, name, imgsz, len(cf), berr, berr / max(H, W), serr))
        else:
            print("end-to-end %s %s@%d: %d boxes within %.1e px / %.1e, %d of them at another position inside a score tie"
                  % (prec, name, imgsz, len(cf), btol, E2E_SCORE, moved))
    else:
        pytest.skip("%d candidates within 1e-5 of the confidence threshold" % near)
