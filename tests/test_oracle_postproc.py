"""Pins oracle/postproc_ref.py against vectors captured from the imported reference."""
import json, os
import numpy as np
import pytest
from oracle import postproc_ref as R

NAMES = {0: "spurious", 1: "compact", 2: "extended", 3: "extended-multisland", 4: "flagged"}


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fp:
        return json.load(fp)


def test_generate_tiles(golden_dir):
    t = load(golden_dir, "tiles.json")
    for name, case in t.items():
        got = R.generate_tiles(*case["args"])
        if case["tiles"] is None:
            assert got is None, name
        else:
            assert [list(x) for x in got] == case["tiles"], name
    assert len(t["c3_16k_512_0.8"]["tiles"]) == 1600
    assert t["c3_16k_512_0.8"]["tiles"][0] == [0, 512, 0, 512]
    assert t["c3_16k_512_0.8"]["tiles"][39] == [15990, 16384, 0, 512]
    assert len(t["c2_16k_512_1.0"]["tiles"]) == 1024
    assert len(t["c5_32k_640_0.8"]["tiles"]) == 4096


def test_neighbor_lists(golden_dir):
    nb = load(golden_dir, "neighbors.json")
    for key, ref in nb.items():
        P = int(key.rsplit("/P", 1)[1])
        grid = [tuple(r["coords"]) for r in ref]
        got = R.create_tile_tasks(grid, P)
        for a, b in zip(got, ref):
            for f in ("wid", "windex", "neighborTaskId", "neighborTaskIndex", "neighborWorkerId"):
                assert a[f] == b[f], (key, a["tid"], f)


def test_process_detections(golden_dir):
    g = np.load(os.path.join(golden_dir, "process_detections.npz"))
    keys = sorted({k.rsplit("/", 1)[0] for k in g.files})
    assert len(keys) == 32
    for k in keys:
        soft, hard, sthr = g[k + "/thr"]
        b, s, c, _ = R.process_detections(g[k + "/in_xyxy"], g[k + "/in_conf"], g[k + "/in_cls"], sthr, soft, hard)
        assert np.array_equal(b, g[k + "/out_xyxy"]), k
        assert np.array_equal(s, g[k + "/out_conf"]), k
        assert np.array_equal(c, g[k + "/out_cls"]), k


def test_serial_catalog(golden_dir):
    fx = load(golden_dir, "catalog_serial.json")
    b, s, c = (np.array(x, np.float32) for x in fx["dets"][0])
    cfg = fx["config"]
    kb, ks, kc, _ = R.process_detections(b, s, c, cfg["score_thr"], cfg["soft"], cfg["hard"])
    objs = R.make_objs(kb, ks, kc, NAMES, 132, 132)
    assert R.catalog_text({"image_id": "galaxy0001", "objs": objs}) == fx["catalog_text"]
    saw = fx["model_saw"][0]
    assert saw["shape"] == [132, 132, 3] and saw["dtype"] == "float64"
    assert saw["kw"]["imgsz"] == 640 and saw["kw"]["conf"] == 0.7 and saw["kw"]["iou"] == 0.5


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_tiled_catalog(golden_dir, tag):
    fx = load(golden_dir, "catalog_tiled_%s.json" % tag)
    grid = [tuple(t) for t in fx["grid"]]
    dets = [tuple(np.array(x, np.float32) for x in d) for d in fx["dets"]]
    skipped = {c[0] for c in fx["calls"] if c[1] < 0}
    ts, cat = R.run_tiled_reference(grid, dets, skipped, NAMES, fx["config"], fx["image_id"])
    assert json.loads(json.dumps(ts)) == fx["tile_sources_before_merge"]["sources"]
    assert R.catalog_text(cat) == fx["catalog_text"]
