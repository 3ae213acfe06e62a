"""Shared machinery of the BASELINE-config parity tests (tests/test_gpu_configs.py) and of tools/fp16_delta.py:
synthetic inputs of the configs at reduced size, the CPU oracle run tile by tile, and detection-set matching.

Reference shapes of the runs: test/run_inference_parallel.sh:18-52 (tiled run), scripts/run.py:254-256 (chan3 needs
nchannels=3), :272-302 (stage order); SURVEY.md section 8(d) "Configs as concrete runs"."""
import numpy as np

CONF, IOU, SOFT, HARD = 0.7, 0.5, 0.3, 0.8
_CACHE = {}

from caesar_yolo_amd.pipelines import SPECS, device_pipeline      # noqa: E402,F401  (the CLI-order stage lists of the configs)

ZS_MINMAX = SPECS["zscale+minmax"]
CHAN3 = SPECS["chan3+minmax"]


def s16k():
    """The benchmark mosaic (caesar_yolo_amd/synth.py, seed 20260104), generated once per test session (1 GiB)."""
    if "s16k" not in _CACHE:
        from caesar_yolo_amd import synth
        _CACHE["s16k"] = synth.make_mosaic(16384, seed=20260104)
    return _CACHE["s16k"]


def config_input(name):
    """-> (image fp32 [H,W], tile size, step fraction, imgsz, stage spec) of a BASELINE config at reduced size."""
    if name == "C1":      # test/galaxy0001.fits as ONE frame (132 x 132 -> LetterBox resize to 640 x 640), test/run_inference.sh:6-10 with BASELINE thresholds
        if "c1" not in _CACHE:
            import os
            from caesar_yolo_amd import utils
            data, _ = utils.read_fits_image(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "galaxy0001.fits"))
            _CACHE["c1"] = np.ascontiguousarray(np.array(data, np.float32))
        return _CACHE["c1"], 132, 1.0, 640, ZS_MINMAX
    if name == "C2":      # stand-in for the missing cutout: crop of S16k at (6144, 6144); 512x512 tiles, step 1.0 -> 16 tiles
        return np.ascontiguousarray(s16k()[6144:8192, 6144:8192]), 512, 1.0, 512, ZS_MINMAX
    if name == "C3":      # crop at (204, 204): tile 2 of rows/columns starts at 1024 = the all-zero block -> two rejected tiles;
        return np.ascontiguousarray(s16k()[204:2252, 204:2252]), 512, 0.8, 512, ZS_MINMAX     # step 0.8 -> 25 tiles incl. ragged
    if name == "C5":      # the S32k recipe (seed 20260105) at 2560x2560: 640x640 tiles, step 0.8 -> 25 tiles, 3-channel preprocessing
        if "c5" not in _CACHE:
            from caesar_yolo_amd import synth
            _CACHE["c5"] = synth.make_mosaic(2560, seed=20260105)
        return _CACHE["c5"], 640, 0.8, 640, CHAN3
    raise KeyError(name)


def oracle_run(name, conf=CONF):
    """CPU oracle over every tile of a config: -> dict(grid, dets [per tile (boxes, scores, classes) after
    process_detections, or None], raw [per tile NMS output before the IoU merge + the kept anchor indices], skipped, near [candidates within 1e-3 of
    the conf cut], catalog)."""
    key = ("oracle", name, conf)
    if key in _CACHE:
        return _CACHE[key]
    from gpu_common import oracle_model
    from oracle import preprocessing_ref as P
    from oracle import postproc_ref as R
    img, ts, step, imgsz, spec = config_input(name)
    grid = R.generate_tiles(0, img.shape[1] - 1, 0, img.shape[0] - 1, ts, ts, step, step)
    dp = P.build_pipeline(spec)
    om = oracle_model()
    dets, raw, skipped, near = [], [], set(), 0
    for tid, (x0, x1, y0, y1) in enumerate(grid):
        tile = np.array(img[y0:y1, x0:x1], np.float32)
        tile[~np.isfinite(tile)] = 0
        im = None if not np.any(tile) else dp(P.to_cube(tile))   # an all-zero tile crashes the reference: status 1 here
        if im is None or P.rows_constant(im):
            skipped.add(tid)
            dets.append(None)
            raw.append(None)
            continue
        det, aidx, _, pred = om.predict_raw(im, imgsz, conf, IOU)
        near += int(((pred[0, 4:].amax(0) - conf).abs() < 1e-3).sum())
        b, s, c = det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy()
        raw.append((b, s, c, aidx.numpy()))
        dets.append(R.process_detections(b, s, c, conf, SOFT, HARD)[:3])
    names = om.names
    thr = {"score_thr": conf, "soft": SOFT, "hard": HARD}
    _, cat = R.run_tiled_reference(grid, [r[:3] if r is not None else (np.zeros((0, 4), np.float32),) * 3 for r in raw],
                                   skipped, names, thr, name)
    out = dict(grid=grid, dets=dets, raw=raw, skipped=skipped, near=near, catalog=cat["sources"], names=names)
    _CACHE[key] = out
    return out


def iou_matrix(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ub = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(ua[:, None] + ub[None, :] - inter, 1e-30)


def match_sets(got, ref, min_iou=0.5):
    """Greedy one-to-one matching (same class, best IoU first) of two detection sets given as (boxes, scores, classes).
    -> dict(matched, missing, extra, max_dbox, max_dscore)."""
    gb, gs, gc = (np.asarray(x) for x in got)
    rb, rs, rc = (np.asarray(x) for x in ref)
    out = dict(matched=0, missing=len(rs), extra=len(gs), max_dbox=0.0, max_dscore=0.0)
    if len(gs) == 0 or len(rs) == 0:
        return out
    m = iou_matrix(gb, rb)
    m[gc.astype(int)[:, None] != rc.astype(int)[None, :]] = 0.0
    used_g, used_r = set(), set()
    for flat in np.argsort(-m, axis=None):
        i, j = divmod(int(flat), m.shape[1])
        if m[i, j] < min_iou:
            break
        if i in used_g or j in used_r:
            continue
        used_g.add(i); used_r.add(j)
        out["max_dbox"] = max(out["max_dbox"], float(np.abs(gb[i] - rb[j]).max()))
        out["max_dscore"] = max(out["max_dscore"], float(abs(gs[i] - rs[j])))
    out["matched"] = len(used_g)
    out["missing"] = len(rs) - len(used_r)
    out["extra"] = len(gs) - len(used_g)
    return out


def sum_reports(reports):
    tot = dict(matched=0, missing=0, extra=0, max_dbox=0.0, max_dscore=0.0)
    for r in reports:
        for k in ("matched", "missing", "extra"):
            tot[k] += r[k]
        for k in ("max_dbox", "max_dscore"):
            tot[k] = max(tot[k], r[k])
    return tot


def sources_as_sets(src):
    b = np.array([[s["x1"], s["y1"], s["x2"], s["y2"]] for s in src], np.float64).reshape(-1, 4)
    return b, np.array([s["score"] for s in src], np.float64), np.array([s["class_id"] for s in src], np.int64)


def nms_level_report(got, ref, H=0, W=0):
    """Kept-box sets after NMS compared by ANCHOR INDEX (the north star's "kept-box index set"): got / ref =
    (boxes, scores, classes, anchors); H, W = letterboxed network input (anchor -> stride).
    -> dict(ref, got, common, max_dbox [px], max_dbox_strides [box delta / the anchor's stride], max_dscore over the common anchors)."""
    gb, gs, gc, ga = (np.asarray(x) for x in got)
    rb, rs, rc, ra = (np.asarray(x) for x in ref)
    gi = {int(a): i for i, a in enumerate(ga)}
    n0, n1 = (H // 8) * (W // 8), (H // 16) * (W // 16)
    out = dict(ref=len(ra), got=len(ga), common=0, class_flips=0, max_dbox=0.0, max_dbox_strides=0.0, max_dscore=0.0)
    for j, a in enumerate(ra):
        i = gi.get(int(a))
        if i is None:
            continue
        out["common"] += 1
        out["class_flips"] += int(int(gc[i]) != int(rc[j]))
        d = float(np.abs(gb[i] - rb[j]).max())
        stride = 8.0 if int(a) < n0 else (16.0 if int(a) < n0 + n1 else 32.0)
        out["max_dbox"] = max(out["max_dbox"], d)
        out["max_dbox_strides"] = max(out["max_dbox_strides"], d / stride)
        out["max_dscore"] = max(out["max_dscore"], float(abs(gs[i] - rs[j])))
    return out


def sum_nms_reports(reports):
    tot = dict(ref=0, got=0, common=0, class_flips=0, max_dbox=0.0, max_dbox_strides=0.0, max_dscore=0.0)
    for r in reports:
        for k in ("ref", "got", "common", "class_flips"):
            tot[k] += r[k]
        for k in ("max_dbox", "max_dbox_strides", "max_dscore"):
            tot[k] = max(tot[k], r[k])
    return tot
