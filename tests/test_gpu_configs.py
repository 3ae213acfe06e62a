"""BASELINE.json configs 2, 3 and 5 on the GPU (config 1: tests/test_gpu_tiled.py; config 4 = config 3 on 8 GPUs, covered
on CPU over gloo in tests/test_distributed_gloo.py and, for the full grid, by the rank-count-invariance property below).

  (a) fp32 context, reduced size, through `SFinder.run_parallel` (the reference's tiled entry, inference.py:578-658):
      catalog vs the CPU oracle run tile by tile -- same sources, classes, edge/merged flags and names; integer
      coordinates equal up to rare +-1 at truncation boundaries;
  (b) fp16 context (what bench.py measures) on the SAME tiles: matched / missing / extra detections, max |dbox| and
      |dscore| against the oracle, printed and asserted against the stated bounds (DESIGN.md section 2 holds the table);
  (c) the full S16k grid (1600 tiles) in the fp16 context as a property test.

Network parity is "unpinned" in the sense of DESIGN.md section 2: the oracle restates the public ultralytics pipeline."""
import json
import os
import numpy as np
import pytest
import torch
from gpu_common import seeded_weights
import config_common as CC

pytestmark = pytest.mark.gpu

# Stated bounds of the fp16 context (fp16 operands, fp32 accumulate; what bench.py measures) against the fp32 oracle, on the
# seeded random-init weights every run here uses.  Three kinds of discontinuity amplify the ~1e-2 logit noise of fp16 operands:
# the conf cut (score > 0.7) and NMS (IoU > 0.5) move single anchors in or out; the IoU merge keeps "the best score of a
# component", so a 3e-3 score wiggle can change WHICH member survives (then the box changes by tens of pixels although every
# logit is within tolerance); and with random weights the DFL bin distributions are broad, so a box edge (= the expectation
# over 16 bins x stride) moves by a fraction of a stride.  Hence: set agreement at every level, score deltas on identical
# anchors, box deltas in units of the anchor's stride.  Measured values: DESIGN.md section 2.
FP16_NMS_MIN_COMMON = 0.95       # kept-anchor sets after NMS: |common| / |oracle kept|
FP16_NMS_MAX_EXTRA = 0.05        # (|fp16 kept| - |common|) / |oracle kept|
FP16_NMS_MAX_DBOX_STRIDES = 0.5  # same anchor in both runs: box delta / stride of the anchor's level (8, 16 or 32 px)
FP16_MAX_DSCORE = 2e-2           # same anchor / matched source
FP16_MIN_MATCHED = 0.95          # after the IoU merge and in the final catalog: same class, IoU >= 0.5
FP16_MAX_EXTRA = 0.05


# bounds of the parity contexts over ALL kept anchors of a config (641 / 1183 / 7500 boxes; the single-image tests assert 5e-3 px):
# the north star's own bar, 1e-4 in normalised image coordinates (0.051 px at 512, 0.064 px at 640), and 2e-5 on scores.
# Measured max |dbox| (PARITYSET lines): fp32 7.1e-3 / 7.3e-3 / 3.5e-2 px, fp16x3 2.0e-3 / 9.5e-3 / 1.6e-2 px
# on C2 / C3 / C5 (the tail of 7500 broad random-init DFL distributions); scores <= 1.3e-5 / 5.8e-6.
PARITY_BOX_NORM, PARITY_SCORE = 1e-4, 2e-5


def _stagewise(name, prec):
    """One BASELINE config at reduced size through the stage entry points (preproc -> forward -> decode/NMS -> IoU merge), per
    shape class.  -> per tile {tid: dict(kept=(boxes, scores, classes, anchors), merged=(boxes, scores, classes))}, letterbox."""
    from caesar_yolo_amd.model import YOLO
    ref = CC.oracle_run(name)
    img, ts, step, imgsz, spec = CC.config_input(name)
    model = YOLO(seeded_weights()[0], precision=prec, max_batch=32, max_imgsz=imgsz, device=0)
    eng = model.engine()
    mosaic = eng.mosaic_to_device(img)
    cfg = CC.device_pipeline(spec).program()
    classes = {}
    for tid, t in enumerate(ref["grid"]):
        classes.setdefault((t[3] - t[2], t[1] - t[0]), []).append(tid)
    out = {}
    for (th, tw), tids in classes.items():
        xy = [(ref["grid"][t][0], ref["grid"][t][2]) for t in tids]
        netin, status, lb = eng.preproc(mosaic, xy, th, tw, imgsz, cfg)
        pred = eng.forward(netin)
        d, anch, cnt = eng.decode_nms(pred, lb.H, lb.W, th, tw, CC.CONF, CC.IOU)
        m, mcnt, _ = eng.iou_merge(d, cnt, CC.CONF, CC.SOFT, CC.HARD)
        torch.cuda.synchronize()
        status, d, anch, cnt, m, mcnt = (x.cpu().numpy() for x in (status, d, anch, cnt, m, mcnt))
        for b, t in enumerate(tids):
            assert (status[b] != 0) == (t in ref["skipped"]), "tile %d rejection differs" % t
            if status[b] != 0:
                continue
            n, k = int(cnt[b]), int(mcnt[b])
            out[t] = dict(kept=(d[b, :n, :4], d[b, :n, 4], d[b, :n, 5], anch[b, :n]), merged=(m[b, :k, :4], m[b, :k, 4], m[b, :k, 5]),
                          HW=(lb.H, lb.W))
    eng.close()
    return out


def _kept_set_report(name, prec):
    """The north star's "kept-box index set after NMS" for a parity context.  Per tile: the anchors the two runs disagree on, and
    whether each of them is a TIE at a cut -- its score within PARITY_SCORE of the conf threshold, or of the score of the last
    box kept when max_det = 300 truncates the list (two anchors whose scores differ by one fp32 ulp swap places there under any
    re-ordering of an fp32 sum, also between two runs of the reference on different CPUs).  Ties are counted and printed; any
    other disagreement fails.  Common anchors: box / score deltas recorded (asserted by the caller), classes equal; the ORDER
    of the kept list must agree up to permutations inside groups of scores closer than PARITY_SCORE."""
    ref = CC.oracle_run(name)
    got = _stagewise(name, prec)
    rep = dict(tiles=0, kept_ref=0, kept_got=0, common=0, ties=0, order_swaps=0, max_dbox=0.0, max_dscore=0.0, tie_tiles=[])
    for t, g in got.items():
        rb, rs, rc, ra = ref["raw"][t]
        gb, gs, gc, ga = g["kept"]
        rep["tiles"] += 1; rep["kept_ref"] += len(ra); rep["kept_got"] += len(ga)
        gi = {int(a): i for i, a in enumerate(ga)}
        ri = {int(a): i for i, a in enumerate(ra)}
        last = min(float(gs[-1]) if len(gs) == 300 else 2.0, float(rs[-1]) if len(rs) == 300 else 2.0)     # the max_det cut, if it binds
        for a in set(gi) ^ set(ri):
            sc = float(gs[gi[a]]) if a in gi else float(rs[ri[a]])
            tie = abs(sc - CC.CONF) <= PARITY_SCORE or abs(sc - last) <= PARITY_SCORE
            assert tie, "%s %s tile %d: anchor %d (score %.7f) is kept by one run only and is no tie (conf %.1f, last kept %.7f)" % (
                name, prec, t, a, sc, CC.CONF, last)
            rep["ties"] += 1
            if t not in rep["tie_tiles"]:
                rep["tie_tiles"].append(t)
        for a, i in gi.items():
            j = ri.get(a)
            if j is None:
                continue
            rep["common"] += 1
            assert int(gc[i]) == int(rc[j]), (name, prec, t, a)
            rep["max_dbox"] = max(rep["max_dbox"], float(np.abs(gb[i] - rb[j]).max()))
            rep["max_dscore"] = max(rep["max_dscore"], float(abs(gs[i] - rs[j])))
            if i != j:
                rep["order_swaps"] += 1
                assert abs(float(rs[j]) - float(rs[min(i, len(rs) - 1)])) <= PARITY_SCORE, "%s %s tile %d: anchor %d at rank %d vs %d is no score tie" % (name, prec, t, a, i, j)
    return rep, got, ref


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C5"])
def test_config_parity_context_kept_anchor_sets(name, prec):
    """Parity bar of the north star on the reduced BASELINE configs, for both parity contexts: kept-anchor set after NMS identical
    to the oracle's (ties at a cut counted, see _kept_set_report), boxes within 1e-4 of the image size and scores within 2e-5 on every kept anchor."""
    rep, _, _ = _kept_set_report(name, prec)
    print("PARITYSET %s" % json.dumps(dict(config=name, precision=prec, **rep)))
    assert rep["kept_ref"] > 100 and rep["common"] >= rep["kept_ref"] - rep["ties"]
    imgsz = CC.config_input(name)[3]
    assert rep["max_dbox"] <= PARITY_BOX_NORM * imgsz and rep["max_dscore"] <= PARITY_SCORE, rep
    assert rep["ties"] <= 4, rep


def _run_sfinder(tmp_path, name, precision, batch=32):
    from caesar_yolo_amd.inference import SFinder
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.config import CONFIG
    from caesar_yolo_amd import utils
    img, ts, step, imgsz, spec = CC.config_input(name)
    path = str(tmp_path / ("%s.fits" % name))
    utils.write_fits_image(path, img)
    model = YOLO(seeded_weights()[0], precision=precision, max_batch=batch, max_imgsz=imgsz, device=0)
    c = dict(CONFIG)
    c.update(image_path=path, preprocess_fcn=CC.device_pipeline(spec), score_thr=CC.CONF, iou_thr=CC.IOU,
             merge_overlap_iou_thr_soft=CC.SOFT, merge_overlap_iou_thr_hard=CC.HARD, devices=["0"], save_region=False,
             img_size=imgsz, split_image_in_tiles=True, tile_xsize=ts, tile_ysize=ts, tile_xstep=step, tile_ystep=step,
             tile_batch=batch, outfile_json=str(tmp_path / ("catalog_%s_%s.json" % (name, precision))))
    sf = SFinder(model, c)
    assert sf.run_parallel() == 0
    got = json.load(open(c["outfile_json"]))["sources"]
    model.engine().close()
    return got, sf.stats


def _compare_exact(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    off = 0
    for g, r in zip(got, ref):
        assert g["class_id"] == r["class_id"] and g["class_name"] == r["class_name"] and g["name"] == r["name"]
        assert g["edge"] == r["edge"] and g["merged"] == r["merged"]
        assert abs(g["score"] - r["score"]) <= 2e-5
        for k in ("x1", "y1", "x2", "y2"):
            d = abs(g[k] - r[k])
            assert d <= 1.0, (g["name"], k, g[k], r[k])
            off += int(d != 0)
    return off


def _compare_matched(got, ref):
    """The coordinate / flag checks of _compare_exact on the sources both catalogs hold (partner = same class, score within 2e-5,
    integer coordinates within one); returns (sources without a partner in either catalog, coordinates off by one)."""
    used, off, lonely = [False] * len(got), 0, 0
    for r in ref:
        for i, g in enumerate(got):
            if used[i] or g["class_id"] != r["class_id"] or abs(g["score"] - r["score"]) > 2e-5:
                continue
            d = [abs(g[k] - r[k]) for k in ("x1", "y1", "x2", "y2")]
            if max(d) <= 1.0:
                used[i] = True
                assert g["class_name"] == r["class_name"] and g["edge"] == r["edge"] and g["merged"] == r["merged"], (g, r)
                off += sum(int(x != 0) for x in d)
                break
        else:
            lonely += 1
    return lonely + used.count(False), off


EXPECT = {  # config: (tiles, rejected tiles, minimum sources in the oracle catalog, minimum cross-tile merged sources)
    "C2": (16, 0, 60, 0),
    "C3": (25, 2, 60, 2),      # tile 12 is the all-zero block (pipeline -> None), tile 17 starts with three all-zero rows (Q1 row check)
    "C5": (25, 0, 60, 3),
}


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
@pytest.mark.parametrize("name", ["C2", "C3", "C5"])
def test_config_catalog_fp32_matches_oracle(name, prec, tmp_path):
    """Both parity contexts (exact fp32; fp16x3 = fp16 high + low halves on the tuned kernels) against the oracle catalog."""
    ref = CC.oracle_run(name)
    ntiles, nrej, min_src, min_merged = EXPECT[name]
    assert len(ref["grid"]) == ntiles and len(ref["skipped"]) == nrej
    assert len(ref["catalog"]) >= min_src, "oracle catalog too small to mean anything: %d" % len(ref["catalog"])
    assert sum(1 for s in ref["catalog"] if s["merged"]) >= min_merged
    got, stats = _run_sfinder(tmp_path, name, prec)
    assert stats["tiles"] == ntiles and stats["skipped"] == nrej
    nref = sum(len(d[1]) for d in ref["dets"] if d is not None)
    if stats["per_tile_detections"] != nref or len(got) != len(ref["catalog"]):
        # the only admissible cause: score ties at a cut (conf threshold / max_det truncation), see _kept_set_report -- which asserts
        # that every disagreement of the kept sets IS such a tie.  A tie moves at most one detection per tied anchor.
        rep, _, _ = _kept_set_report(name, prec)
        assert rep["ties"] > 0, "catalog differs from the oracle's although the kept-anchor sets agree"
        cat = CC.match_sets(CC.sources_as_sets(got), CC.sources_as_sets(ref["catalog"]))
        print("%s %s: %d score ties at a cut (tiles %s) -> per-tile detections %d vs %d, catalog %s" % (
            name, prec, rep["ties"], rep["tie_tiles"], stats["per_tile_detections"], nref, cat))
        assert abs(stats["per_tile_detections"] - nref) <= rep["ties"]
        assert cat["missing"] + cat["extra"] <= 2 * rep["ties"] and cat["max_dscore"] <= PARITY_SCORE
        lonely, off = _compare_matched(got, ref["catalog"])          # every other source: same flags, integer coordinates within one
        assert lonely <= 2 * rep["ties"] and off <= max(2, len(got) // 25), (lonely, off)
        return
    off = _compare_exact(got, ref["catalog"])
    print("%s %s: %d sources (%d merged across tiles), %d of %d integer coordinates differ by one" % (
        name, prec, len(got), sum(1 for s in got if s["merged"]), off, 4 * len(got)))
    assert off <= max(2, len(got) // 25)


@pytest.mark.parametrize("name", ["C2", "C3", "C5"])
def test_config_fp16_delta_vs_oracle(name, tmp_path):
    """The benchmarked mode against the oracle on the same tiles, at three levels: kept anchors after NMS, per-tile
    detections after the IoU merge, final catalog.  Prints one FP16DELTA line (tools/fp16_delta.py collects them)."""
    ref = CC.oracle_run(name)
    nms_reports, reports, nref = [], [], 0
    for t, g in _stagewise(name, "fp16").items():
        nms_reports.append(CC.nms_level_report(g["kept"], ref["raw"][t], *g["HW"]))
        reports.append(CC.match_sets(g["merged"], ref["dets"][t]))
        nref += len(ref["dets"][t][1])
    nms_rep, tile_rep = CC.sum_nms_reports(nms_reports), CC.sum_reports(reports)
    cat16, _ = _run_sfinder(tmp_path, name, "fp16")
    cat_rep = CC.match_sets(CC.sources_as_sets(cat16), CC.sources_as_sets(ref["catalog"]))
    print("FP16DELTA %s" % json.dumps({"config": name, "nms": nms_rep, "oracle_per_tile_detections": nref, "per_tile": tile_rep,
                                       "oracle_sources": len(ref["catalog"]), "catalog": cat_rep,
                                       "candidates_within_1e-3_of_conf": ref["near"]}))
    assert nms_rep["ref"] > 0 and nms_rep["common"] >= FP16_NMS_MIN_COMMON * nms_rep["ref"], nms_rep
    assert nms_rep["got"] - nms_rep["common"] <= max(2, FP16_NMS_MAX_EXTRA * nms_rep["ref"]), nms_rep
    assert nms_rep["max_dbox_strides"] <= FP16_NMS_MAX_DBOX_STRIDES and nms_rep["max_dscore"] <= FP16_MAX_DSCORE, nms_rep
    for rep, n in ((tile_rep, nref), (cat_rep, len(ref["catalog"]))):
        assert n > 0
        assert rep["matched"] >= FP16_MIN_MATCHED * n, rep
        assert rep["extra"] <= max(2, FP16_MAX_EXTRA * n), rep
        assert rep["max_dscore"] <= FP16_MAX_DSCORE, rep


def test_s16k_full_grid_fp16_properties(monkeypatch):
    """BASELINE config 3 at full size (16384^2, 1600 tiles) in the benchmarked fp16 context, through size-independent
    properties: every tile reports a status (no tile of this grid is rejected: neither the NaN strip nor the all-zero
    block covers a whole tile or its first rows), the catalog does not depend on the batch size or on the split forward (= it would not depend on the
    rank count, cf. tests/test_distributed_gloo.py), boxes lie inside the mosaic, names are S1..SN in order."""
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.inference import TileEngine
    from caesar_yolo_amd import utils
    monkeypatch.delenv("CY_BATCH_INVARIANT", raising=False)
    img = CC.s16k()
    grid = utils.generate_tiles(0, 16383, 0, 16383, 512, 512, 0.8, 0.8)
    assert len(grid) == 1600
    model = YOLO(seeded_weights()[0], precision="fp16", max_batch=256, max_imgsz=512, device=0)
    det = model.engine()
    mosaic = det.mosaic_to_device(img)
    cfg = CC.device_pipeline(CC.ZS_MINMAX).program()
    cats = {}
    for batch, dual in ((256, "1"), (64, "0")):
        monkeypatch.setenv("CY_DUAL_FORWARD", dual)
        eng = TileEngine(det, mosaic, grid, cfg, 512, CC.CONF, CC.IOU, CC.SOFT, CC.HARD, 0, 1, batch)
        eng.run_local()
        eng.gather()
        torch.cuda.synchronize()
        src, stats = eng.catalog(model.names)
        cats[(batch, dual)] = (src, stats)
    (src, stats), (src2, stats2) = cats[(256, "1")], cats[(64, "0")]
    print("S16k fp16: %d per-tile detections, %d sources, %d tiles skipped" % (stats["per_tile_detections"], len(src), stats["skipped"]))
    assert stats["tiles"] == 1600 and stats["skipped"] == stats2["skipped"] == 0
    assert stats["per_tile_detections"] == stats2["per_tile_detections"] and src == src2
    assert 15000 <= stats["per_tile_detections"] <= 21000 and 8000 <= len(src) <= 10500   # seeded:l:5 at conf 0.7 (18097 / 9202 on this build)
    assert [s["name"] for s in src] == ["S%d" % (i + 1) for i in range(len(src))]
    b = np.array([[s["x1"], s["y1"], s["x2"], s["y2"]] for s in src])
    assert b.min() >= 0 and b[:, [0, 2]].max() <= 16384 and b[:, [1, 3]].max() <= 16384
    assert (b[:, 2] >= b[:, 0]).all() and (b[:, 3] >= b[:, 1]).all()
    assert sum(1 for s in src if s["merged"]) >= 500                                 # the overlap bands merge thousands of edge sources
    det.close()
