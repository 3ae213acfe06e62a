"""Whole-path parity on the GPU (f32 context) through the reference-shaped host surface:
  * BASELINE config 1: test/galaxy0001.fits (copied to tests/golden as a data fixture), single frame, zscale+minmax,
    imgsz 640, scoreThr 0.7, iouThr 0.5  -> SFinder.run() -> out_galaxy0001.json
  * a tiled run (golden mosaic "c": 900x700, 256x256 tiles, step 0.5 -> 48 tiles incl. ragged and rejected ones)
    -> SFinder.run_parallel() -> catalog_*.json
against the CPU oracle run tile by tile (numpy preprocessing -> torch fp32 network -> NMS -> process_detections ->
make_objs -> flag_edge_sources -> merge_edge_sources).  Catalog integers come from int() truncation of fp32 boxes, so a
box within 2e-3 px of an integer may legitimately differ by one; such cases are counted and must be rare."""
import json
import os
import numpy as np
import pytest
import torch
from gpu_common import seeded_weights, oracle_model, assert_same_detections, ROOT

pytestmark = pytest.mark.gpu
CONF, IOU, SOFT, HARD = 0.7, 0.5, 0.3, 0.8


def _oracle_tile(img2d, imgsz, conf=CONF):
    from oracle import preprocessing_ref as P
    from oracle import postproc_ref as R
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    tile = np.array(img2d, np.float32)
    tile[~np.isfinite(tile)] = 0
    im = dp(P.to_cube(tile))
    if im is None or P.rows_constant(im):
        return None
    det, _, _, _ = oracle_model().predict_raw(im, imgsz, conf, IOU)
    return R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), conf, SOFT, HARD)[:3]


def _config(path, **kw):
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import preprocessing as PP
    c = dict(CONFIG)
    c.update(image_path=path, preprocess_fcn=PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]),
             score_thr=CONF, iou_thr=IOU, merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD, devices=["0"],
             save_region=False)
    c.update(kw)
    return c


def _compare_sources(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    off_by_one = 0
    for g, r in zip(got, ref):
        assert g["class_id"] == r["class_id"] and g["class_name"] == r["class_name"]
        assert g["edge"] == r["edge"] and g.get("merged") == r.get("merged") and g["name"] == r["name"]
        assert abs(g["score"] - r["score"]) <= 2e-5
        for k in ("x1", "y1", "x2", "y2"):
            d = abs(g[k] - r[k])
            assert d <= 1.0, (k, g[k], r[k])
            off_by_one += int(d != 0)
    return off_by_one


def test_config1_single_frame_catalog(tmp_path, monkeypatch):
    from caesar_yolo_amd.inference import SFinder
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import utils
    from oracle import postproc_ref as R
    monkeypatch.chdir(tmp_path)
    path = os.path.join(ROOT, "tests/golden/galaxy0001.fits")
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=1, max_imgsz=640, device=0)
    sf = SFinder(model, _config(path, img_size=640))
    assert sf.run() == 0
    got = json.load(open(tmp_path / "out_galaxy0001.json"))
    data, _ = utils.read_fits_image(path)
    kb, ks, kc = _oracle_tile(np.asarray(data, np.float32), 640)
    ref = {"image_id": "galaxy0001", "objs": R.make_objs(kb, ks, kc, model.names, 132, 132)}
    assert got["image_id"] == ref["image_id"]
    n1 = _compare_sources(got["objs"], ref["objs"])
    assert n1 <= max(1, len(ref["objs"]) // 50), "%d of %d catalog coordinates differ by one" % (n1, 4 * len(ref["objs"]))


def test_tiled_catalog_matches_oracle(tmp_path, monkeypatch):
    from caesar_yolo_amd.inference import SFinder
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import utils
    from oracle import postproc_ref as R
    monkeypatch.chdir(tmp_path)
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"]
    img = img.copy()
    img[10:40, 300:330] = np.nan                       # non-finite pixels -> 0 on ingest (utils.py:219, :394)
    path = str(tmp_path / "mosaic_c.fits")
    utils.write_fits_image(path, img)
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=16, max_imgsz=256, device=0)
    sf = SFinder(model, _config(path, img_size=256, split_image_in_tiles=True, tile_xsize=256, tile_ysize=256,
                                tile_xstep=0.5, tile_ystep=0.5, tile_batch=16))
    assert sf.run_parallel() == 0
    got = json.load(open(tmp_path / "catalog_mosaic_c.json"))["sources"]
    grid = utils.generate_tiles(0, img.shape[1] - 1, 0, img.shape[0] - 1, 256, 256, 0.5, 0.5)
    assert len(grid) == 48
    dets, skipped = [], set()
    for tid, (x0, x1, y0, y1) in enumerate(grid):
        r = _oracle_tile(img[y0:y1, x0:x1], 256)
        if r is None:
            skipped.add(tid)
            dets.append(None)
        else:
            dets.append(r)
    assert sf.stats["skipped"] == len(skipped) and len(skipped) >= 3
    tasks = R.create_tile_tasks(grid, 1)
    tile_sources = []
    for t in tasks:
        tid = t["tid"]
        if tid in skipped or len(dets[tid][1]) == 0:
            continue
        x0, x1, y0, y1 = t["coords"]
        objs = R.make_objs(dets[tid][0], dets[tid][1], dets[tid][2], model.names, x1 - x0, y1 - y0, x0, y0, "t%d" % tid)
        R.flag_edge_sources(objs, t["coords"], [tasks[k]["coords"] for k in t["neighborTaskId"]])
        tile_sources.append({"objs": objs, "tileId": tid, "neighborTileIds": t["neighborTaskId"]})
    ref = R.merge_edge_sources(tile_sources)
    n1 = _compare_sources(got, ref)
    assert n1 <= max(2, len(ref) // 25), "%d catalog coordinates differ by one" % n1


def test_predict_tiles_batched_entry():
    """YOLO.predict_tiles (SURVEY §8b additive entry): per-tile Results equal the oracle's process_detections output, and
    a tile whose pipeline gives None (all-zero after ingest) comes back as None."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import preprocessing as PP
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"].astype(np.float32)
    img[0:256, 512:768] = 0.0
    model = YOLO(seeded_weights()[0], precision="fp32", max_batch=8, max_imgsz=256, device=0)
    eng = model.engine()
    mosaic = eng.mosaic_to_device(img)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    coords = [(0, 256, 0, 256), (256, 512, 0, 256), (512, 768, 0, 256), (128, 384, 128, 384), (600, 856, 400, 656)]
    conf = 0.3                                         # nothing on these tiles reaches the 0.7 of the catalog tests
    res = model.predict_tiles(mosaic, coords, cfg, imgsz=256, conf=conf, iou=IOU,
                              merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD)
    assert len(res) == len(coords) and res[2] is None
    ndet = 0
    for (x0, x1, y0, y1), r in zip(coords, res):
        ref = _oracle_tile(img[y0:y1, x0:x1], 256, conf)
        if ref is None:
            assert r is None
            continue
        kb, ks, kc = ref
        assert len(r.boxes.conf) == len(ks)
        ndet += len(ks)
        if len(ks):
            assert_same_detections(r.boxes.xyxy.cpu().numpy(), r.boxes.conf.cpu().numpy(), r.boxes.cls.cpu().numpy(), kb, ks, kc,
                                   5e-3, 2e-5, "tile (%d, %d)" % (x0, y0))          # pixels; scores
    assert ndet >= 8
    with pytest.raises(ValueError):
        model.predict_tiles(mosaic, [(0, 256, 0, 256), (0, 200, 0, 256)], cfg, imgsz=256)
    assert model.predict_tiles(mosaic, [], cfg) == []


def test_split_forward_gives_the_same_detections(monkeypatch):
    """cy_detect_tiles runs batches of 64..239 tiles as two half-batches on two streams (second workspace); forced here on
    a batch of 5 (3 + 2): same per-tile detections as the single-stream forward (fp16 context)."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import preprocessing as PP
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"].astype(np.float32)
    model = YOLO(seeded_weights()[0], precision="fp16", max_batch=8, max_imgsz=256, device=0)
    eng = model.engine()
    mosaic = eng.mosaic_to_device(img)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    coords = [(0, 256, 0, 256), (256, 512, 0, 256), (512, 768, 0, 256), (128, 384, 128, 384), (600, 856, 400, 656)]
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CY_DUAL_FORWARD", mode)
        res = model.predict_tiles(mosaic, coords, cfg, imgsz=256, conf=0.3, iou=IOU,      # (nothing reaches 0.7 on these tiles)
                                  merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD)
        out[mode] = [None if r is None else (r.boxes.xyxy.cpu().numpy(), r.boxes.conf.cpu().numpy(), r.boxes.cls.cpu().numpy())
                     for r in res]
    assert sum(len(o[1]) for o in out["0"] if o is not None) > 0
    for a, b in zip(out["0"], out["2"]):
        assert (a is None) == (b is None)
        if a is None:
            continue
        assert len(a[1]) == len(b[1])
        np.testing.assert_allclose(a[1], b[1], atol=5e-3)
        np.testing.assert_array_equal(a[2], b[2])
        np.testing.assert_allclose(a[0], b[0], atol=0.5)


def test_batch_invariant_detections_do_not_depend_on_batching(monkeypatch):
    """With the default (batch-invariant) kernel selection the per-tile detections are the same bits whether the tiles go through cy_detect_tiles
    one by one, all together, or as two half-batches on two streams."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import preprocessing as PP
    monkeypatch.delenv("CY_BATCH_INVARIANT", raising=False)          # the default selection is the invariant one
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"].astype(np.float32)
    model = YOLO(seeded_weights()[0], precision="fp16", max_batch=8, max_imgsz=256, device=0)
    mosaic = model.engine().mosaic_to_device(img)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    coords = [(0, 256, 0, 256), (256, 512, 0, 256), (128, 384, 128, 384), (600, 856, 400, 656), (300, 556, 200, 456)]
    kw = dict(imgsz=256, conf=0.3, iou=IOU, merge_overlap_iou_thr_soft=SOFT, merge_overlap_iou_thr_hard=HARD)

    def run(groups, mode):
        monkeypatch.setenv("CY_DUAL_FORWARD", mode)
        out = []
        for g in groups:
            out += model.predict_tiles(mosaic, g, cfg, **kw)
        return [None if r is None else np.concatenate([r.boxes.xyxy.cpu().numpy(), r.boxes.conf.cpu().numpy()[:, None],
                                                       r.boxes.cls.cpu().numpy()[:, None]], 1) for r in out]
    together = run([coords], "0")
    alone = run([[c] for c in coords], "0")
    split = run([coords], "2")
    assert sum(len(t) for t in together if t is not None) >= 8
    for a, b, c in zip(together, alone, split):
        assert (a is None) == (b is None) == (c is None)
        if a is not None:
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(a, c)


def test_small_batch_lane_and_fence(monkeypatch):
    """cy_detect_tiles' small-batch lane (batches of at most a third of max_batch on their own stream / buffer set / workspace) gives
    the same bits as the main lane (CY_SMALL_LANE is read per call), and the ordering contract of unflushed calls holds: a memset
    of an output buffer queued on the caller's stream BETWEEN two unflushed calls is ordered before the next call's internal
    streams by cy_detect_fence (include/caesar_yolo_hip.h), so the results land after it."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd import preprocessing as PP
    monkeypatch.delenv("CY_BATCH_INVARIANT", raising=False)
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_c.npz"))["img"].astype(np.float32)
    model = YOLO(seeded_weights()[0], precision="fp16", max_batch=12, max_imgsz=256, device=0)
    det = model.engine()
    mosaic = det.mosaic_to_device(img)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    big = [(64 * i, 40 * (i % 3)) for i in range(9)]            # 9 tiles: main lane (9 * 3 > 12)
    small = [(600, 400), (300, 200), (128, 128)]                 # 3 tiles: small lane (3 * 3 <= 12)
    args = (256, 256, 256, cfg, 0.3, IOU, SOFT, HARD)

    def run(lane, fenced):
        monkeypatch.setenv("CY_SMALL_LANE", lane)
        seq = (big, small, big, small)
        # every output buffer exists and is filled BEFORE the first call (the contract for work that is not fenced)
        outs = [(torch.full((len(xy), 300, 6), 7.0, device="cuda"), torch.full((len(xy),), 77, dtype=torch.int32, device="cuda"),
                 torch.full((len(xy),), 7, dtype=torch.int32, device="cuda")) for xy in seq]
        torch.cuda.synchronize()
        for k, (xy, o) in enumerate(zip(seq, outs)):
            if fenced and k > 0:
                for t in o:
                    t.zero_()                                    # queued on the caller's stream between unflushed calls
                det.fence()
            det.detect_tiles(mosaic, xy, *args, out=o, flush=False)
        det.flush()
        torch.cuda.synchronize()
        return [(d.cpu().numpy(), c.cpu().numpy(), s.cpu().numpy()) for d, c, s in outs]
    ref = run("0", False)
    assert sum(int(c.sum()) for _, c, _ in ref) >= 8 and all((c != 77).all() and (s != 7).all() and (s == 0).any() for _, c, s in ref)
    for lane, fenced in (("1", False), ("1", True), ("0", True)):
        got = run(lane, fenced)
        for (d0, c0, s0), (d1, c1, s1) in zip(ref, got):
            np.testing.assert_array_equal(c0, c1)
            np.testing.assert_array_equal(s0, s1)
            for b in range(len(c0)):
                np.testing.assert_array_equal(d0[b, :c0[b]], d1[b, :c0[b]])
    det.close()


def test_compact_records_matches_numpy():
    """cy_compact_records (rank 0 after the gather): counts (0 for rejected tiles), statuses, exclusive prefix, total, and the valid
    detections in tile-id order -- against the same selection in numpy, for a permuted tile order, 3 ranks, more tiles than one
    scan round of 1024 threads, and tiles with 0 and 300 detections."""
    from caesar_yolo_amd.model import HipDetector
    import caesar_yolo_amd.lib as L
    det = HipDetector(seeded_weights()[0], device=0, precision="fp16", max_batch=1, max_imgsz=64)
    rng = np.random.default_rng(11)
    R, rows, T = 3, 700, 2050                               # 2100 row slots, 2050 tiles
    g = rng.standard_normal((R, rows, 1803)).astype(np.float32)
    cnt = rng.integers(0, 301, size=(R, rows)).astype(np.float32)
    cnt.reshape(-1)[:5] = [0, 300, 0, 1, 300]
    st = (rng.random((R, rows)) < 0.1).astype(np.float32) * rng.integers(1, 3, size=(R, rows))
    g[:, :, -3], g[:, :, -2] = cnt, st
    perm = rng.permutation(R * rows)[:T].astype(np.int64)
    gd, pd = torch.from_numpy(g).cuda(), torch.from_numpy(perm).cuda()
    hdr = torch.full((3 * T + 1,), -1, dtype=torch.int32, device="cuda")
    out = torch.zeros((T * 300 * 6,), dtype=torch.float32, device="cuda")
    det.compact_records(gd, pd, hdr, out)
    torch.cuda.synchronize()
    h = hdr.cpu().numpy()
    flat = g.reshape(-1, 1803)[perm]
    st_ref = flat[:, -2].astype(np.int64)
    cnt_ref = np.where(st_ref == 0, flat[:, -3].astype(np.int64), 0)
    np.testing.assert_array_equal(h[:T], cnt_ref)
    np.testing.assert_array_equal(h[T:2 * T], st_ref)
    np.testing.assert_array_equal(h[2 * T:3 * T], np.concatenate([[0], np.cumsum(cnt_ref)[:-1]]))
    assert h[3 * T] == cnt_ref.sum()
    ref = np.concatenate([flat[t, :cnt_ref[t] * 6] for t in range(T)])
    np.testing.assert_array_equal(out.cpu().numpy()[:ref.size], ref)
    # row indices outside the gathered buffer: those tiles come back rejected (status CY_ERR_ARG, count 0), nothing is read
    bad = perm.copy()
    bad[[3, 1500]] = [R * rows, -7]
    det.compact_records(gd, torch.from_numpy(bad).cuda(), hdr, out)
    torch.cuda.synchronize()
    h2 = hdr.cpu().numpy()
    cnt2, st2 = cnt_ref.copy(), st_ref.copy()
    cnt2[[3, 1500]], st2[[3, 1500]] = 0, -1          # CY_ERR_ARG
    np.testing.assert_array_equal(h2[:T], cnt2)
    np.testing.assert_array_equal(h2[T:2 * T], st2)
    assert h2[3 * T] == cnt2.sum()
    det.close()
