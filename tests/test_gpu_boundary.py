"""The drop-in boundary driven the way the reference drives it:
  * the model call with EXACTLY the keyword set of caesar_yolo/evaluation.py:181-193 (incl. device='cpu', the reference's
    default `--devices=cpu`, scripts/run.py:130) and the result access of :261-265;
  * `Analyzer(model, config).predict(image, image_id, header, xmin, ymin)` (:128-245): return codes, `.results`,
    tile-origin offsets, the object name tag, rejection of None / constant-row images."""
import json
import os
import numpy as np
import pytest
from gpu_common import seeded_weights, oracle_model, assert_same_detections, ROOT

pytestmark = pytest.mark.gpu
CONF, IOU, SOFT, HARD = 0.7, 0.5, 0.3, 0.8


def _galaxy_cube():
    from caesar_yolo_amd import utils
    from oracle import preprocessing_ref as P
    data, _ = utils.read_fits_image(os.path.join(ROOT, "tests/golden/galaxy0001.fits"))
    img = np.asarray(data, np.float32)
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    return img, dp(P.to_cube(img))


def test_model_call_with_the_reference_keyword_set(caplog):
    from caesar_yolo_amd.model import YOLO
    img, cube = _galaxy_cube()
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=1, max_imgsz=640)
    assert isinstance(model.names, dict) and len(model.names) == 5
    # caesar_yolo/evaluation.py:181-193, verbatim keyword set; the image is the (H,W,3) float64 cube in [0,255]
    results = model(cube, save=False, device="cpu", imgsz=640, conf=CONF, iou=IOU, visualize=False, show=False,
                    show_labels=False, show_conf=False, show_boxes=False)
    got = []
    for result in results:                                       # :261-265
        bboxes = result.boxes.xyxy.cpu().numpy()
        scores = result.boxes.conf.cpu().numpy()
        labels = result.boxes.cls.cpu().numpy()
        got.append((bboxes, scores, labels))
        for c in labels:
            assert int(c) in model.names
    assert len(got) == 1
    det, _, _, _ = oracle_model().predict_raw(cube, 640, CONF, IOU)
    b, s, c = got[0]
    assert b.dtype == np.float32 and b.shape == (len(s), 4) and len(s) == det.shape[0] and len(s) > 0
    # scores 2e-5 (measured <= 7e-6), boxes 5e-3 px (measured <= 2.1e-3), classes equal; rows inside a score tie may swap
    moved = assert_same_detections(b, s, c, det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), 5e-3, 2e-5, "model call")
    print("model call, reference keyword set: %d boxes, %d inside score ties at another position" % (len(s), moved))
    assert (np.diff(s) <= 0).all() and b.min() >= 0 and b.max() <= 132          # conf-descending, clipped to the image


def test_analyzer_predict_contract(tmp_path, monkeypatch):
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.evaluation import Analyzer
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import preprocessing as PP
    from oracle import postproc_ref as R
    monkeypatch.chdir(tmp_path)
    img, cube = _galaxy_cube()
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=1, max_imgsz=640)
    c = dict(CONFIG)
    c.update(img_size=640, preprocess_fcn=PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]),
             score_thr=CONF, iou_thr=IOU, merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD, devices=["cpu"])
    an = Analyzer(model, c)
    an.obj_name_tag = "t7"
    assert an.predict(img, image_id="galaxy0001", header=None, xmin=1000, ymin=2000) == 0
    det, _, _, _ = oracle_model().predict_raw(cube, 640, CONF, IOU)
    kb, ks, kc, _ = R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), CONF, SOFT, HARD)
    ref = R.make_objs(kb, ks, kc, model.names, 132, 132, 1000, 2000, "t7")
    assert len(ref) > 0 and len(an.bboxes_final) == len(an.scores_final) == len(an.class_ids_final) == len(an.labels_final) == len(ref)
    assert an.results["image_id"] == "galaxy0001" and len(an.results["objs"]) == len(ref)
    for g, r in zip(an.results["objs"], ref):
        assert set(g) == set(r) == {"name", "x1", "x2", "y1", "y2", "class_id", "class_name", "score", "edge"}
        assert g["name"] == r["name"] and g["name"].endswith("_t7")
        assert g["class_id"] == r["class_id"] and g["class_name"] == r["class_name"] and g["edge"] == r["edge"]
        assert abs(g["score"] - r["score"]) <= 2e-5
        for k in ("x1", "x2", "y1", "y2"):
            assert abs(g[k] - r[k]) <= 1.0 and g[k] >= 1000 - 1                  # tile origin added (evaluation.py:460-463)
    assert json.load(open(tmp_path / "out_galaxy0001.json")) == json.loads(json.dumps(an.results))
    # rejections: a None image, a pipeline that returns None (all zero), constant first rows (Q1) -> -1, nothing written
    assert an.predict(None) == -1
    assert Analyzer(model, c).predict(np.zeros((132, 132), np.float32), image_id="z") == -1
    q = img.copy()
    q[1, :] = 0.0
    assert Analyzer(model, c).predict(q, image_id="q") == -1
    assert not os.path.exists(tmp_path / "out_z.json") and not os.path.exists(tmp_path / "out_q.json")


def test_counters_report_candidate_overflow_and_degenerate_boxes():
    """A context with a deliberately small candidate capacity counts the tiles that overflow it (cy_detect_counters); the
    default capacity (= anchor count) cannot overflow even at conf 0."""
    from caesar_yolo_amd.model import HipDetector
    from caesar_yolo_amd import preprocessing as PP
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = np.ascontiguousarray(g["in/syn192"])
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    small = HipDetector(seeded_weights()[0], device=0, precision="fp16", max_batch=2, max_imgsz=192, max_cand=64)
    mosaic = small.mosaic_to_device(img)
    small.counters(reset=True)
    small.detect_tiles(mosaic, [(0, 0), (0, 0)], 192, 192, 192, cfg, 0.001, IOU, SOFT, HARD)
    assert small.counters(reset=True)["cand_overflow_tiles"] == 2
    assert small.counters()["cand_overflow_tiles"] == 0
    small.close()
    full = HipDetector(seeded_weights()[0], device=0, precision="fp16", max_batch=2, max_imgsz=192)
    mosaic = full.mosaic_to_device(img)
    d, cnt, st = full.detect_tiles(mosaic, [(0, 0)], 192, 192, 192, cfg, 0.0, IOU, SOFT, HARD)
    assert full.counters()["cand_overflow_tiles"] == 0 and int(cnt[0]) > 0
    full.close()


def test_analyzer_plot_and_preprocessed_fits(tmp_path, monkeypatch):
    """--draw_plots --save_plots / save_img (caesar_yolo/evaluation.py:203-210, :237-243, :351-411, :550-554): a PNG of the
    preprocessed image with the final boxes, and channel 0 of the preprocessed image as a float64 FITS equal to the
    reference-pinned oracle's output."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.evaluation import Analyzer
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import preprocessing as PP, utils
    monkeypatch.chdir(tmp_path)
    img, cube = _galaxy_cube()
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=1, max_imgsz=640)
    c = dict(CONFIG)
    c.update(img_size=640, preprocess_fcn=PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]),
             score_thr=CONF, iou_thr=IOU, merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD, devices=["0"],
             draw_plot=True, save_plot=True, draw_class_label_in_caption=True, save_img=True, save_region=True)
    an = Analyzer(model, c)
    assert an.predict(img, image_id="galaxy0001") == 0 and len(an.bboxes_final) > 0
    assert os.path.getsize(tmp_path / "out_galaxy0001.png") > 10000
    assert an.image.shape == (132, 132, 3)
    np.testing.assert_allclose(an.image, cube, rtol=1e-10, atol=1e-12)
    data, hdr = utils.get_fits_header(str(tmp_path / "out_galaxy0001.fits"))[0], None
    assert data["BITPIX"] == -64 and data["NAXIS1"] == 132 and data["NAXIS2"] == 132
    raw = np.memmap(str(tmp_path / "out_galaxy0001.fits"), dtype=">f8", mode="r", offset=2880, shape=(132, 132))
    np.testing.assert_allclose(np.asarray(raw), cube[:, :, 0], rtol=1e-10, atol=1e-12)
    reg = open(tmp_path / "out_galaxy0001.reg").read().splitlines()
    assert reg[1] == "image" and len(reg) == 2 + len(an.bboxes_final)


def test_analyzer_predict_three_channel_array():
    """Analyzer.predict on a caller-supplied (H,W,3) array whose channels DIFFER (caesar_yolo/evaluation.py:146-154 takes a 3-D image as
    is): each plane through its channel's program on the device, the cube through the model call -- against the oracle pipeline on
    the same cube.  Also: a cube whose first rows are constant is rejected by the row check (:171-176)."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.evaluation import Analyzer
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import preprocessing as PP
    from oracle import preprocessing_ref as P
    from oracle import postproc_ref as R
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    base = np.asarray(g["in/big512"][:256, :256], np.float32)
    cube_in = np.stack([base, base[::-1, :] * 1.5 + 0.001, np.roll(base, 17, axis=1)], axis=2).astype(np.float32)
    for prec in ("fp32", "fp16x3"):
        model = YOLO(seeded_weights()[0], precision=prec, max_batch=1, max_imgsz=256)
        for spec, dpre in (([("zscale", dict(contrasts=[0.25, 0.3, 0.4])), ("minmax", dict(norm_min=0, norm_max=255))],
                            PP.DataPreprocessor([PP.ZScaleTransformer([0.25, 0.3, 0.4]), PP.MinMaxNormalizer(0, 255)])),):
            c = dict(CONFIG)
            c.update(img_size=256, preprocess_fcn=dpre, score_thr=0.05, iou_thr=IOU, merge_overlap_iou_thr_soft=SOFT,
                     merge_overlap_iou_thr_hard=HARD, devices=["0"])
            an = Analyzer(model, c)
            assert an.predict(cube_in, image_id="cube") == 0
            ref_img = P.build_pipeline(spec)(cube_in.astype(np.float64))
            det, _, _, _ = oracle_model().predict_raw(ref_img, 256, 0.05, IOU)        # (a cube of unrelated channels scores low on the seeded weights)
            kb, ks, kc, _ = R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), 0.05, SOFT, HARD)
            assert len(an.scores_final) == len(ks) and len(ks) >= 3
            assert_same_detections(an.bboxes_final, an.scores_final, an.class_ids_final, kb, ks, kc, 5e-3, 2e-5, "Analyzer " + prec)
        flat = cube_in.copy()
        flat[:2] = 0.0                                            # rows 0 and 1 all zero in every channel -> constant rows
        assert Analyzer(model, c).predict(flat, image_id="flat") == -1
        assert Analyzer(model, c).predict(np.zeros((64, 64, 3), np.float32), image_id="zero") == -1       # pipeline returns None
        model.engine().close()


def test_serial_run_with_sub_image_ranges(tmp_path, monkeypatch):
    """SFinder.run with --xmin/--xmax/--ymin/--ymax (caesar_yolo/inference.py:499-505 -> utils.read_fits_crop, utils.py:340-394): the crop
    [ymin:ymax, xmin:xmax] (max excluded) is what the model sees, catalog coordinates are relative to it; the reference's argument
    errors return -1."""
    from caesar_yolo_amd.inference import SFinder
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import preprocessing as PP, utils
    from oracle import preprocessing_ref as P
    from oracle import postproc_ref as R
    monkeypatch.chdir(tmp_path)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = np.asarray(g["in/big512"], np.float32)
    path = str(tmp_path / "big.fits")
    utils.write_fits_image(path, img)
    model = YOLO(seeded_weights()[0], precision="fp16x3", max_batch=1, max_imgsz=256)

    def cfg(**kw):
        c = dict(CONFIG)
        c.update(image_path=path, img_size=256, score_thr=0.3, iou_thr=IOU, merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD,
                 preprocess_fcn=PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]), devices=["0"],
                 save_catalog=True, save_region=False)
        c.update(kw)
        return c
    sf = SFinder(model, cfg(image_xmin=100, image_xmax=340, image_ymin=37, image_ymax=293))
    assert sf.run() == 0
    got = json.load(open(tmp_path / "out_big.json"))["objs"]
    crop = img[37:293, 100:340]
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    det, _, _, _ = oracle_model().predict_raw(dp(P.to_cube(crop)), 256, 0.3, IOU)
    kb, ks, kc, _ = R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), 0.3, SOFT, HARD)
    ref = R.make_objs(kb, ks, kc, model.names, 240, 256)
    assert len(got) == len(ref) and len(ref) >= 3
    for a, b in zip(got, ref):
        assert a["class_id"] == b["class_id"] and abs(a["score"] - b["score"]) <= 2e-5
        assert all(abs(a[k] - b[k]) <= 1 for k in ("x1", "y1", "x2", "y2"))
        assert 0 <= a["x1"] and a["x2"] <= 240 and a["y2"] <= 256              # relative to the crop
    for bad in (dict(image_xmin=-1, image_xmax=100, image_ymin=0, image_ymax=50), dict(image_xmin=50, image_xmax=50, image_ymin=0, image_ymax=50),
                dict(image_xmin=10, image_xmax=60, image_ymin=80, image_ymax=20)):
        assert SFinder(model, cfg(**bad)).run() == -1
    model.engine().close()
