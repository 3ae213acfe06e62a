"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol the header declares,
and its host-only entry points (graph plan, letterbox geometry, catalog records, cross-tile merge) agree with
the Python spec / the oracle / the golden vectors.  No compute kernels are launched here."""
import ctypes as C
import json, os, re
import numpy as np
import pytest
from caesar_yolo_amd import lib as L
from caesar_yolo_amd import yolov8_spec as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "spurious", 1: "compact", 2: "extended", 3: "extended-multisland", 4: "flagged"}


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "caesar_yolo_hip.h")).read()
    declared = set(re.findall(r"\b(cy_[a-z0-9_]+)\s*\(", hdr))
    lib = L.load()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert declared == set(L.EXPORTS)


@pytest.mark.parametrize("scale,nc", [("l", 5), ("l", 80), ("n", 5), ("s", 3), ("m", 5)])
def test_plan_matches_python_spec(scale, nc):
    got = L.plan_convs(scale, nc)
    want = [(c.name, c.cin, c.cout, c.k, c.s, int(c.act)) for c in S.conv_list(scale, nc)]
    assert got == want


def test_published_model_size():
    # ultralytics publishes 43.7 M params / 165.2 GFLOPs for yolov8l (nc=80, 640x640)
    total, _ = S.conv_macs("l", 80, 640, 640)
    params = sum(c.cout * c.cin * c.k * c.k + c.cout for c in S.conv_list("l", 80))
    assert abs(params / 1e6 - 43.67) < 0.05
    assert abs(2 * total / 1e9 - 165.1) < 0.3
    assert S.conv_macs("l", 5, 512, 512)[0] == 52743667712 or abs(S.conv_macs("l", 5, 512, 512)[0] / 1e9 - 52.744) < 1e-3


def test_letterbox_geometry_matches_oracle():
    from oracle import yolov8_ref as Y
    for (h, w, s) in [(132, 132, 640), (512, 512, 512), (512, 394, 512), (394, 512, 512), (394, 394, 512),
                      (256, 256, 640), (640, 512, 640), (100, 333, 640), (700, 300, 512), (33, 47, 64)]:
        nh, nw, top, bottom, left, right, H, W = Y.letterbox_params(h, w, s)
        lb = L.letterbox(h, w, s)
        assert (lb.new_h, lb.new_w, lb.top, lb.left, lb.H, lb.W) == (nh, nw, top, left, H, W), (h, w, s)
    assert L.load().cy_num_anchors(512, 512) == 5376 and L.load().cy_num_anchors(640, 640) == 8400
    assert L.load().cy_num_anchors(512, 416) == 4368


def _records_to_sources(rec, n, names):
    out = []
    for i in range(n):
        x1, y1, x2, y2, score, cid, e, m = rec[i]
        edge = {0.0: 0, 1.0: 1, 2.0: True}[e]
        out.append({"name": "S%d" % (i + 1), "x1": x1, "x2": x2, "y1": y1, "y2": y2, "class_id": int(cid),
                    "class_name": names[int(cid)], "score": score, "edge": edge, "merged": bool(m)})
    return out


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_catalog_host_functions_against_reference_catalogs(golden_dir, tag):
    """cy_make_tile_records + cy_merge_edge_sources reproduce the reference's tiled catalog byte for byte when fed the
    per-tile merged detections (computed here by the oracle's process_detections on the recorded fake-model output)."""
    from oracle import postproc_ref as R
    fx = json.load(open(os.path.join(golden_dir, "catalog_tiled_%s.json" % tag)))
    grid = np.array(fx["grid"], np.int32)
    skipped = {c[0] for c in fx["calls"] if c[1] < 0}
    dets, dtile = [], []
    for tid, d in enumerate(fx["dets"]):
        if tid in skipped:
            continue
        b, s, c = (np.array(x, np.float32) for x in d)
        kb, ks, kc, _ = R.process_detections(b, s, c, fx["config"]["score_thr"], fx["config"]["soft"], fx["config"]["hard"])
        for i in range(len(ks)):
            dets.append([kb[i][0], kb[i][1], kb[i][2], kb[i][3], ks[i], float(kc[i])])
            dtile.append(tid)
    det = np.ascontiguousarray(np.array(dets, np.float32).reshape(-1, 6))
    dtile = np.ascontiguousarray(np.array(dtile, np.int32))
    n, T = len(dtile), len(grid)
    rec = np.zeros((n, 8), np.float64)
    lib = L.load()
    ip, fp, dp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.check(lib.cy_make_tile_records(det.ctypes.data_as(fp), dtile.ctypes.data_as(ip), n, grid.ctypes.data_as(ip), T,
                                     rec.ctypes.data_as(dp)))
    # per-tile objects before the merge, including the int/True edge tri-state
    ref_objs = [o for t in fx["tile_sources_before_merge"]["sources"] for o in t["objs"]]
    assert len(ref_objs) == n
    for i, o in enumerate(ref_objs):
        e = {0.0: 0, 1.0: 1, 2.0: True}[rec[i, 7]]
        assert (rec[i, 0], rec[i, 1], rec[i, 2], rec[i, 3]) == (o["x1"], o["y1"], o["x2"], o["y2"]), i
        assert e == o["edge"] and type(e) == type(o["edge"]), (i, e, o["edge"])
        assert rec[i, 4] == o["score"] and int(rec[i, 5]) == o["class_id"]
    out = np.zeros((n, 8), np.float64)
    m = L.check(lib.cy_merge_edge_sources(rec.ctypes.data_as(dp), n, grid.ctypes.data_as(ip), T, out.ctypes.data_as(dp)))
    text = json.dumps({"sources": _records_to_sources(out, m, NAMES)}, indent=2, sort_keys=True)
    assert text == fx["catalog_text"]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.CyError):
        L.load()


def test_merge_is_independent_of_thread_count():
    """cy_merge_edge_sources splits its pair search over host threads (CY_MERGE_THREADS, read once per process): same
    catalog for 1, 4 and 16 threads on 40k synthetic detections (16 used to overrun the per-thread pair lists)."""
    import subprocess, sys, hashlib
    code = r'''
import sys, hashlib
import numpy as np
sys.path.insert(0, %r)
from caesar_yolo_amd import utils
from caesar_yolo_amd.inference import merge_records
grid = utils.generate_tiles(0, 8191, 0, 8191, 512, 512, 0.8, 0.8)
rng = np.random.default_rng(7)
det, dt = [], []
for t, (x0, x1, y0, y1) in enumerate(grid):
    n = 100
    nx, ny = x1 - x0, y1 - y0
    cx, cy = rng.uniform(0, nx, n), rng.uniform(0, ny, n)
    w, h = rng.uniform(4, 40, n), rng.uniform(4, 40, n)
    b = np.stack([np.clip(cx - w, 0, nx), np.clip(cy - h, 0, ny), np.clip(cx + w, 0, nx), np.clip(cy + h, 0, ny),
                  rng.uniform(0.7, 1.0, n), rng.integers(0, 5, n)], 1)
    det.append(b); dt.append(np.full(n, t))
det = np.ascontiguousarray(np.concatenate(det), np.float32); dt = np.ascontiguousarray(np.concatenate(dt), np.int32)
out = merge_records(det, dt, grid)
print(len(det), out.shape[0], hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest())
''' % ROOT
    res = set()
    for nt in ("1", "4", "16"):
        env = dict(os.environ, CY_MERGE_THREADS=nt)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.add(r.stdout.strip())
    assert len(res) == 1, res
    n_in, n_out = [int(v) for v in res.pop().split()[:2]]
    assert n_in >= 40000 and 0 < n_out < n_in


def test_package_import_raises_the_hip_queue_count():
    """Importing the package (before any HIP call) asks the HIP runtime for eight hardware queues: a context's four streams plus a
    collective library's otherwise share four (profiles/r04_stream_budget.md).  An explicit setting wins."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, sys; sys.path.insert(0, %r); import caesar_yolo_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip() == "4"
