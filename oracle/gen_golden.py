#!/opt/conda/bin/python3.9
"""Golden-vector generator.  TEST INFRASTRUCTURE ONLY -- runs only in the build
container (it imports the read-only reference at /root/reference); nothing in the
product path, the GPU tests, smoke() or bench.py imports or executes this file.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore \
        /root/repo/oracle/gen_golden.py /root/repo/tests/golden

It drives the reference's own host code (caesar_yolo.preprocessing / utils / graph /
evaluation / inference) with a replayable fake model (the network itself lives in
ultralytics, which is absent from the container -- SURVEY.md section 8c) and stores
inputs + outputs as small fixtures.  Only data is written: arrays and JSON.

Stubs (no arithmetic in any of them): regions, cv2, numpyencoder, fitsio (a thin
functional shim over astropy.io.fits), plus numpy alias shims needed by astropy 4.3.1
on numpy 1.26.  See SURVEY.md Appendix D.
"""
import sys, os, types, json, copy, io, contextlib, logging
import numpy as np

# ---------------------------------------------------------------- numpy shims
for _n, _v in (("float", float), ("int", int), ("bool", bool), ("object", object),
               ("complex", complex), ("str", str)):
    if not hasattr(np, _n):
        setattr(np, _n, _v)
if not hasattr(np, "asscalar"):
    np.asscalar = lambda a: a.item()
if not hasattr(np, "alen"):
    np.alen = len

import matplotlib
matplotlib.use("Agg")
from astropy.io import fits as _afits


# ---------------------------------------------------------------- module stubs
def _install_stubs():
    reg = types.ModuleType("regions")

    class _Any(object):
        def __init__(self, *a, **k):
            self.a, self.k = a, k

        def write(self, *a, **k):
            pass
    for n in ("RegionMeta", "RegionVisual", "RectanglePixelRegion", "PolygonPixelRegion",
              "PixCoord", "Regions"):
        setattr(reg, n, type(n, (_Any,), {}))
    reg.write_ds9 = lambda *a, **k: None
    sys.modules["regions"] = reg
    sys.modules["cv2"] = types.ModuleType("cv2")

    npe = types.ModuleType("numpyencoder")

    class NumpyEncoder(json.JSONEncoder):
        def default(self, o):
            if isinstance(o, np.integer):
                return int(o)
            if isinstance(o, np.floating):
                return float(o)
            if isinstance(o, np.bool_):
                return bool(o)
            if isinstance(o, np.ndarray):
                return o.tolist()
            return json.JSONEncoder.default(self, o)
    npe.NumpyEncoder = NumpyEncoder
    sys.modules["numpyencoder"] = npe

    fio = types.ModuleType("fitsio")

    class _HDU(object):
        def __init__(self, hdu):
            self.hdu = hdu

        def get_dims(self):
            return list(self.hdu.data.shape)

        def __getitem__(self, sl):
            d = self.hdu.data[sl]
            return np.array(d, dtype=d.dtype.newbyteorder("="))

        def read_header(self):
            return self.hdu.header

    class FITS(object):
        def __init__(self, fn):
            self.h = _afits.open(fn, memmap=False)

        def __getitem__(self, i):
            return _HDU(self.h[i])

        def close(self):
            self.h.close()
    fio.FITS = FITS
    fio.FITSHDR = dict
    sys.modules["fitsio"] = fio


_install_stubs()
sys.path.insert(0, "/root/reference")
import caesar_yolo  # noqa: E402
from caesar_yolo import utils as rutils  # noqa: E402
from caesar_yolo import preprocessing as rpp  # noqa: E402
from caesar_yolo.graph import Graph  # noqa: E402
from caesar_yolo.evaluation import Analyzer  # noqa: E402
from caesar_yolo.inference import SFinder  # noqa: E402
from caesar_yolo.config import CONFIG as RCONFIG  # noqa: E402
from astropy.visualization import ZScaleInterval  # noqa: E402
from astropy.stats import sigma_clip, sigma_clipped_stats  # noqa: E402

caesar_yolo.logger.setLevel(logging.ERROR)
NAMES = {0: "spurious", 1: "compact", 2: "extended", 3: "extended-multisland", 4: "flagged"}


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# ---------------------------------------------------------------- synthetic tiles
def make_tile(h, w, seed, nsrc=25, zero_block=True, zero_strip=True):
    """Seeded radio-like tile (fp32), post-read semantics: non-finite already -> 0."""
    rng = np.random.default_rng(seed)
    img = rng.normal(0.0, 1e-4, (h, w))
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nsrc):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sm = rng.uniform(1.5, 6.0)
        q = rng.uniform(0.5, 1.0)
        pa = rng.uniform(0, np.pi)
        pk = 10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2))
        dx, dy = xx - cx, yy - cy
        u = dx * np.cos(pa) + dy * np.sin(pa)
        v = -dx * np.sin(pa) + dy * np.cos(pa)
        img += pk * np.exp(-0.5 * ((u / sm) ** 2 + (v / (sm * q)) ** 2))
    img = img.astype(np.float32)
    if zero_strip:
        img[:, : max(2, w // 20)] = 0.0      # a NaN border strip after read -> 0
    if zero_block:
        img[h // 2: h // 2 + h // 8, w // 3: w // 3 + w // 8] = 0.0
    return img


def cube(img):
    """Analyzer.predict 2D -> (H,W,3) float64 replicate (evaluation.py:146-154)."""
    c = np.zeros((img.shape[0], img.shape[1], 3))
    for i in range(3):
        c[:, :, i] = img
    return c


PIPELINES = {
    "bkg": lambda: [rpp.BkgSubtractor(sigma=3)],
    "bkg_box": lambda: [rpp.BkgSubtractor(sigma=3, use_mask_box=True, mask_fract=0.7)],
    "shift": lambda: [rpp.SigmaClipShifter(sigma=1)],
    "clip": lambda: [rpp.SigmaClipper(sigma_low=10, sigma_up=10)],
    "clip_1_3": lambda: [rpp.SigmaClipper(sigma_low=1, sigma_up=3)],
    "zscale": lambda: [rpp.ZScaleTransformer(contrasts=[0.25, 0.25, 0.25])],
    "zscale_c40": lambda: [rpp.ZScaleTransformer(contrasts=[0.4, 0.4, 0.4])],
    "minmax": lambda: [rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "zscale_minmax": lambda: [rpp.ZScaleTransformer(contrasts=[0.25, 0.25, 0.25]),
                              rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "chan3_minmax": lambda: [rpp.ChanResizer(nchans=3),
                             rpp.Chan3Trasformer(sigma_clip_baseline=0, sigma_clip_low=10,
                                                 sigma_clip_up=10, zscale_contrast=0.25),
                             rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "chan3_1_20": lambda: [rpp.ChanResizer(nchans=3),
                           rpp.Chan3Trasformer(sigma_clip_baseline=0, sigma_clip_low=1,
                                               sigma_clip_up=20, zscale_contrast=0.25)],
    "full": lambda: [rpp.BkgSubtractor(sigma=3), rpp.SigmaClipShifter(sigma=1),
                     rpp.SigmaClipper(sigma_low=10, sigma_up=10),
                     rpp.ZScaleTransformer(contrasts=[0.25, 0.25, 0.25]),
                     rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
}


def gen_preproc(outdir):
    gal = _afits.open("/root/reference/test/galaxy0001.fits", memmap=False)[0].data
    gal = np.array(gal, dtype=np.float32)
    gal[~np.isfinite(gal)] = 0
    inputs = {
        "galaxy": gal,
        "syn192": make_tile(192, 192, 11),
        "rag": make_tile(200, 160, 12, nsrc=12),
        "dense": make_tile(96, 128, 13, nsrc=30, zero_block=False, zero_strip=False),
    }
    out = {}
    for iname, img in inputs.items():
        out["in/" + iname] = img
        for pname, mk in PIPELINES.items():
            dp = rpp.DataPreprocessor(mk())
            with quiet():
                res = dp(cube(img))
            assert res is not None
            if pname.startswith("chan3"):
                out["out/%s/%s" % (iname, pname)] = res
            else:
                assert np.array_equal(res[:, :, 0], res[:, :, 1])
                out["out/%s/%s" % (iname, pname)] = res[:, :, 0]
        # raw statistics the device kernels must reproduce
        d = img.astype(np.float64)
        nz = d[np.logical_and(d != 0, np.isfinite(d))]
        zl = {}
        for c in (0.25, 0.4):
            vmin, vmax = ZScaleInterval(contrast=c).get_limits(d)
            zl[str(c)] = [float(vmin), float(vmax)]
        out["stats/%s/zscale_0.25" % iname] = np.array(zl["0.25"])
        out["stats/%s/zscale_0.4" % iname] = np.array(zl["0.4"])
        for (lo, up) in ((10, 10), (1, 3), (0, 10), (1, 20), (3, 3), (1, 1)):
            r = sigma_clip(nz, sigma_lower=lo, sigma_upper=up, masked=True, return_bounds=True)
            out["stats/%s/sigclip_bounds_%g_%g" % (iname, lo, up)] = np.array([float(r[1]), float(r[2])])
        for s in (3, 1):
            m = sigma_clipped_stats(nz, sigma=s)
            out["stats/%s/sigstats_%g" % (iname, s)] = np.array([float(x) for x in m])
    # one full-size 512x512 tile: statistics + strided pixel sample only (keeps the file small)
    big = make_tile(512, 512, 14, nsrc=60)
    out["in/big512"] = big
    for pname in ("zscale_minmax", "chan3_minmax", "full"):
        dp = rpp.DataPreprocessor(PIPELINES[pname]())
        with quiet():
            res = dp(cube(big))
        flat = res.reshape(-1, 3)
        out["big512_sample/%s" % pname] = flat[::97].copy()
    d = big.astype(np.float64)
    nz = d[d != 0]
    out["stats/big512/zscale_0.25"] = np.array([float(x) for x in ZScaleInterval(contrast=0.25).get_limits(d)])
    r = sigma_clip(nz, sigma_lower=10, sigma_upper=10, masked=True, return_bounds=True)
    out["stats/big512/sigclip_bounds_10_10"] = np.array([float(r[1]), float(r[2])])
    out["stats/big512/sigstats_3"] = np.array([float(x) for x in sigma_clipped_stats(nz, sigma=3)])
    # degenerate: MinMaxNormalizer on an all-zero tile returns None (preprocessing.py:101-103)
    z = np.zeros((16, 16), np.float32)
    with quiet():
        assert rpp.DataPreprocessor(PIPELINES["zscale_minmax"]())(cube(z)) is None
    np.savez_compressed(os.path.join(outdir, "preproc.npz"), **out)
    print("preproc.npz keys:", len(out))


CHID_PIPELINES = {
    # --bkg_chid / --clip_chid (scripts/run.py:89, :98, :275-281): the stage touches ONE channel, the others pass through
    "bkg_chid0": lambda: [rpp.BkgSubtractor(sigma=3, chid=0)],
    "shiftclip_chid1_minmax": lambda: [rpp.SigmaClipShifter(sigma=1, chid=1), rpp.SigmaClipper(sigma_low=10, sigma_up=10, chid=1),
                                       rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "bkg_chid2_zscale_minmax": lambda: [rpp.BkgSubtractor(sigma=3, chid=2), rpp.ZScaleTransformer(contrasts=[0.25, 0.25, 0.25]),
                                        rpp.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "bkgbox_chid1_clip_chid0": lambda: [rpp.BkgSubtractor(sigma=3, use_mask_box=True, mask_fract=0.7, chid=1),
                                        rpp.SigmaClipper(sigma_low=1, sigma_up=3, chid=0)],
}


def gen_preproc_chid(outdir):
    """Per-channel stage selection on the inputs preproc.npz already holds (same arrays): full (H,W,3) outputs."""
    g = np.load(os.path.join(outdir, "preproc.npz"))
    out = {}
    for iname in ("galaxy", "syn192", "rag"):
        img = g["in/" + iname]
        for pname, mk in CHID_PIPELINES.items():
            with quiet():
                res = rpp.DataPreprocessor(mk())(cube(img))
            assert res is not None and res.shape == img.shape + (3,)
            out["out/%s/%s" % (iname, pname)] = res
    np.savez_compressed(os.path.join(outdir, "preproc_chid.npz"), **out)
    print("preproc_chid.npz keys:", len(out))


# ---------------------------------------------------------------- tiles + neighbours
TILE_CASES = {
    "c2_16k_512_1.0": (0, 16383, 0, 16383, 512, 512, 1.0, 1.0),
    "c3_16k_512_0.8": (0, 16383, 0, 16383, 512, 512, 0.8, 0.8),
    "c5_32k_640_0.8": (0, 32767, 0, 32767, 640, 640, 0.8, 0.8),
    "sliver_1300x900_512_0.8": (0, 1299, 0, 899, 512, 512, 0.8, 0.8),
    "s2048_512_1.0": (0, 2047, 0, 2047, 512, 512, 1.0, 1.0),
    "s2500_512_0.8": (0, 2499, 0, 2499, 512, 512, 0.8, 0.8),
    "rect_700x1500_256x384_0.5_0.9": (0, 699, 0, 1499, 256, 384, 0.5, 0.9),
    "exact_1024_512_0.5": (0, 1023, 0, 1023, 512, 512, 0.5, 0.5),
    "par_132_64_1.0": (0, 131, 0, 131, 64, 64, 1.0, 1.0),
}


class FakeMPI(object):
    """config['mpi'] stand-in with the methods SFinder calls (inference.py:557-572, 1086-1109)."""

    class _Comm(object):
        def __init__(self, size, rank):
            self.size, self.rank = size, rank

        def Get_size(self):
            return self.size

        def Get_rank(self):
            return self.rank

        def Barrier(self):
            pass

        def Get_group(self):
            return None

        def Create_group(self, g, tag):
            return self

        def send(self, *a, **k):
            raise RuntimeError("no p2p in the harness")

        def recv(self, *a, **k):
            raise RuntimeError("no p2p in the harness")

    class Group(object):
        @staticmethod
        def Incl(g, ids):
            return type("G", (), {"size": len(ids)})()

    def __init__(self, size=1, rank=0):
        self.COMM_WORLD = FakeMPI._Comm(size, rank)

    def Get_version(self):
        return (3, 1)


def write_fits(path, data, hdr_extra=None):
    h = _afits.Header()
    h["BMAJ"] = 0.0026
    h["BMIN"] = 0.0021
    h["BPA"] = 84.0
    h["CDELT1"] = -0.0005
    h["CDELT2"] = 0.0005
    _afits.PrimaryHDU(np.asarray(data, dtype=np.float32), h).writeto(path, overwrite=True)


def base_config(image_path, **kw):
    cfg = copy.deepcopy({k: v for k, v in RCONFIG.items() if k != "mpi"})
    cfg["mpi"] = None
    cfg["image_path"] = image_path
    cfg["image_xmin"] = cfg["image_xmax"] = cfg["image_ymin"] = cfg["image_ymax"] = -1
    cfg["max_ntasks_per_worker"] = 100000
    cfg["save_region"] = False
    cfg.update(kw)
    return cfg


def gen_tiles(outdir, scratch):
    out = {}
    for name, a in TILE_CASES.items():
        with quiet():
            g = rutils.generate_tiles(*a)
        out[name] = {"args": list(a), "tiles": [list(map(int, t)) for t in g]}
    # rejected argument sets (utils.py:626-645) -> None
    rej = {
        "tile_gt_image": (0, 99, 0, 99, 512, 512, 1.0, 1.0),
        "step_gt_1": (0, 999, 0, 999, 100, 100, 1.2, 1.0),
        "step_zero": (0, 999, 0, 999, 100, 100, 0.0, 1.0),
        "xmax_le_xmin": (5, 5, 0, 99, 10, 10, 1.0, 1.0),
    }
    for name, a in rej.items():
        with quiet():
            assert rutils.generate_tiles(*a) is None
        out["rejected/" + name] = {"args": list(a), "tiles": None}
    with open(os.path.join(outdir, "tiles.json"), "w") as fp:
        json.dump(out, fp)

    # neighbour lists from SFinder.create_tile_tasks at P in {1,2,4,8}
    nb = {}
    for name in ("sliver_1300x900_512_0.8", "s2048_512_1.0", "s2500_512_0.8",
                 "rect_700x1500_256x384_0.5_0.9", "exact_1024_512_0.5"):
        a = TILE_CASES[name]
        nx, ny = a[1] + 1, a[3] + 1
        path = os.path.join(scratch, "nb_%s.fits" % name)
        write_fits(path, np.zeros((ny, nx), np.float32))
        for P in (1, 2, 4, 8):
            cfg = base_config(path, split_image_in_tiles=True, tile_xsize=a[4], tile_ysize=a[5],
                              tile_xstep=a[6], tile_ystep=a[7])
            cfg["mpi"] = FakeMPI(P, 0)
            sf = SFinder(model=type("M", (), {"names": NAMES})(), config=cfg)
            with quiet():
                sf.init_mpi()
                assert sf.set_img_size_params() == 0
                assert sf.create_tile_tasks() == 0
            per_tid = {}
            for w, tasks in enumerate(sf.tasks_per_worker):
                for j, t in enumerate(tasks):
                    per_tid[int(t.tid)] = {
                        "wid": int(t.wid), "windex": j,
                        "coords": [int(t.ix_min), int(t.ix_max), int(t.iy_min), int(t.iy_max)],
                        "neighborTaskId": [int(x) for x in t.neighborTaskId],
                        "neighborTaskIndex": [int(x) for x in t.neighborTaskIndex],
                        "neighborWorkerId": [int(x) for x in t.neighborWorkerId],
                    }
            nb["%s/P%d" % (name, P)] = [per_tid[k] for k in sorted(per_tid)]
    with open(os.path.join(outdir, "neighbors.json"), "w") as fp:
        json.dump(nb, fp)
    print("tiles.json / neighbors.json written")


# ---------------------------------------------------------------- fake model
class _T(object):
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class FakeResult(object):
    def __init__(self, xyxy, conf, cls):
        self.boxes = type("B", (), {})()
        self.boxes.xyxy = _T(np.asarray(xyxy, np.float32).reshape(-1, 4))
        self.boxes.conf = _T(np.asarray(conf, np.float32).reshape(-1))
        self.boxes.cls = _T(np.asarray(cls, np.float32).reshape(-1))


class ReplayModel(object):
    """Fake detector: call k returns the k-th prepared detection set; records what it saw."""
    names = NAMES

    def __init__(self, dets):
        self.dets, self.k, self.seen = dets, 0, []

    def __call__(self, img, **kw):
        self.seen.append({"shape": list(img.shape), "dtype": str(img.dtype),
                          "min": float(img.min()), "max": float(img.max()),
                          "kw": {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in kw.items()}})
        d = self.dets[self.k]
        self.k += 1
        return [FakeResult(*d)]


def rand_boxes(rng, n, w, h, cluster=0.5, smin=0.5, tie=False):
    """Random boxes with deliberate overlaps: a fraction are jittered copies of earlier ones."""
    xyxy, conf, cls = [], [], []
    for i in range(n):
        if i > 0 and rng.uniform() < cluster:
            j = int(rng.integers(0, i))
            b = np.array(xyxy[j]) + rng.normal(0, 3.0, 4)
            c = cls[j] if rng.uniform() < 0.6 else int(rng.integers(0, 5))
        else:
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            bw, bh = rng.uniform(4, 60), rng.uniform(4, 60)
            b = np.array([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2])
            c = int(rng.integers(0, 5))
        b[0], b[2] = np.clip(sorted((b[0], b[2])), 0, w)
        b[1], b[3] = np.clip(sorted((b[1], b[3])), 0, h)
        if b[2] - b[0] < 1 or b[3] - b[1] < 1:
            b = np.array([10.0, 10.0, 20.0, 20.0])
        xyxy.append(b.astype(np.float32))
        conf.append(np.float32(rng.uniform(smin, 1.0)))
        cls.append(c)
    conf = np.array(conf, np.float32)
    if tie and n >= 4:
        conf[1::3] = conf[0]                      # exact score ties -> first-wins rule matters
    order = np.argsort(-conf, kind="stable")     # ultralytics returns conf-descending
    return (np.array(xyxy, np.float32).reshape(-1, 4)[order], conf[order],
            np.array(cls, np.float32)[order])


def gen_process_detections(outdir):
    rng = np.random.default_rng(2026)
    cases = {}
    specs = [("n0", 0, 0.5, False), ("n1", 1, 0.5, False), ("n5", 5, 0.8, False), ("n40", 40, 0.6, False),
             ("n40_ties", 40, 0.6, True), ("n120", 120, 0.7, True), ("n300", 300, 0.5, False),
             ("n300_dense", 300, 0.9, True)]
    out = {}
    for name, n, cl, tie in specs:
        for (soft, hard, sthr) in ((0.3, 0.8, 0.5), (0.3, 0.8, 0.7), (0.5, 0.6, 0.5), (0.05, 0.1, 0.5)):
            b, s, c = rand_boxes(rng, n, 512, 512, cluster=cl, smin=0.45, tie=tie)
            cfg = base_config("x.fits", merge_overlap_iou_thr_soft=soft, merge_overlap_iou_thr_hard=hard,
                              score_thr=sthr)
            an = Analyzer(ReplayModel([]), cfg)
            with quiet():
                assert an.process_detections([FakeResult(b, s, c)]) == 0
            key = "%s/%g_%g_%g" % (name, soft, hard, sthr)
            out[key + "/in_xyxy"], out[key + "/in_conf"], out[key + "/in_cls"] = b, s, c
            out[key + "/thr"] = np.array([soft, hard, sthr])
            out[key + "/out_xyxy"] = np.array(an.bboxes_final, np.float32).reshape(-1, 4)
            out[key + "/out_conf"] = np.array(an.scores_final, np.float32)
            out[key + "/out_cls"] = np.array(an.class_ids_final, np.int32)
            # which input rows survived (index into the score-filtered list)
            cases[key] = len(an.bboxes_final)
    np.savez_compressed(os.path.join(outdir, "process_detections.npz"), **out)
    print("process_detections.npz cases:", len(cases))


# ---------------------------------------------------------------- catalogs (serial + tiled)
def run_serial(outdir, scratch):
    """C1-shaped run: galaxy0001.fits, zscale+minmax(0,255), imgsz 640, fake detections."""
    os.chdir(scratch)
    rng = np.random.default_rng(7)
    dets = [rand_boxes(rng, 12, 132, 132, cluster=0.6, smin=0.6)]
    # push two boxes onto the frame border to exercise the edge rule (evaluation.py:452-456)
    dets[0][0][0] = np.array([0.4, 20.2, 30.7, 50.9], np.float32)
    dets[0][0][1] = np.array([100.2, 90.5, 131.6, 131.2], np.float32)
    model = ReplayModel(dets)
    dp = rpp.DataPreprocessor(PIPELINES["zscale_minmax"]())
    cfg = base_config("/root/reference/test/galaxy0001.fits", preprocess_fcn=dp, img_size=640,
                      score_thr=0.7, iou_thr=0.5, devices=["cpu"])
    sf = SFinder(model, cfg)
    with quiet():
        assert sf.run() == 0
    with open(os.path.join(scratch, "out_galaxy0001.json")) as fp:
        cat_text = fp.read()
    fix = {"dets": [[d[0].tolist(), d[1].tolist(), d[2].tolist()] for d in dets],
           "model_saw": model.seen, "catalog_text": cat_text,
           "config": {"score_thr": 0.7, "iou_thr": 0.5, "soft": 0.3, "hard": 0.8, "img_size": 640}}
    with open(os.path.join(outdir, "catalog_serial.json"), "w") as fp:
        json.dump(fix, fp)
    print("catalog_serial.json: model saw", model.seen[0]["shape"], model.seen[0]["min"], model.seen[0]["max"])


def dense_edge_dets(rng, grid, dets, nlong=14):
    """Sources that cross tile borders on purpose: long thin rectangles in MOSAIC coordinates, cut by every tile they
    touch (each piece ends exactly on a tile border, so it is edge-flagged and overlaps its neighbours' pieces) -> merge
    chains over three and more tiles; every other one is given twins of EQUAL area in neighbouring tiles (ties in the
    largest-area selection, inference.py:849-851: first wins) and alternating classes (the cross-tile merge ignores class)."""
    nx = max(t[1] for t in grid)
    ny = max(t[3] for t in grid)
    for k in range(nlong):
        horizontal = (k % 2 == 0)
        if horizontal:      # bands in the upper half, 60 px apart: chains do not touch each other
            y0 = 12.0 + 60.0 * (k // 2) + float(rng.uniform(0, 8)); h = float(rng.uniform(8, 22))
            x0 = float(rng.uniform(0, nx * 0.3)); x1 = float(rng.uniform(nx * 0.6, nx - 1))
            gx0, gx1, gy0, gy1 = x0, x1, y0, y0 + h
        else:               # columns in the lower half, 90 px apart
            x0 = 20.0 + 90.0 * (k // 2) + float(rng.uniform(0, 8)); w = float(rng.uniform(8, 22))
            y0 = float(rng.uniform(ny * 0.56, ny * 0.62)); y1 = float(rng.uniform(ny * 0.9, ny - 1))
            gx0, gx1, gy0, gy1 = x0, x0 + w, y0, y1
        score = float(rng.uniform(0.75, 0.99))
        for ti, t in enumerate(grid):
            cx0, cx1 = max(gx0, t[0]), min(gx1, t[1])
            cy0, cy1 = max(gy0, t[2]), min(gy1, t[3])
            if cx1 - cx0 < 4 or cy1 - cy0 < 4:
                continue
            b, sc, c = dets[ti]
            box = [cx0 - t[0], cy0 - t[2], cx1 - t[0], cy1 - t[2]]
            b = np.concatenate([b, np.array([box], np.float32)]) if len(b) else np.array([box], np.float32)
            # equal scores in every piece for odd k (ties), tile-dependent scores otherwise
            sc = np.concatenate([sc, np.array([score if k % 4 < 2 else max(0.71, score - 0.01 * (ti % 7))], np.float32)])
            c = np.concatenate([c, np.array([float((k + ti) % 5)], np.float32)])
            dets[ti] = (b, sc, c)
    return dets


def run_tiled(outdir, scratch, tag, nx, ny, tsize, step, seed, nper=10, dense=False, save_img=True):
    os.chdir(scratch)
    img = make_tile(ny, nx, seed, nsrc=80, zero_block=False, zero_strip=False)
    # tile-skip triggers: one all-zero tile (MinMaxNormalizer -> None) and one tile whose first
    # three rows are constant (evaluation.py:171-176 indexes rows)
    with quiet():
        grid = rutils.generate_tiles(0, nx - 1, 0, ny - 1, tsize, tsize, step, step)
    T = len(grid)
    tz = grid[min(3, T - 1)]
    img[tz[2]:tz[3], tz[0]:tz[1]] = 0.0
    tq = grid[min(5, T - 1)]
    img[tq[2]:tq[2] + 3, tq[0]:tq[1]] = 0.0
    path = os.path.join(scratch, "mosaic_%s.fits" % tag)
    write_fits(path, img)
    rng = np.random.default_rng(seed + 1000)
    dets = []
    for t in grid:
        w, h = t[1] - t[0], t[3] - t[2]
        b, s, c = rand_boxes(rng, nper, w, h, cluster=0.4, smin=0.55)
        # force some boxes onto tile borders / into overlap regions
        for k in range(min(4, len(b))):
            side = int(rng.integers(0, 4))
            if side == 0:
                b[k][0] = rng.uniform(0, 0.9)
            elif side == 1:
                b[k][2] = w - rng.uniform(0, 0.9)
            elif side == 2:
                b[k][1] = rng.uniform(0, 0.9)
            else:
                b[k][3] = h - rng.uniform(0, 0.9)
        dets.append((b, s, c))
    if dense:
        dets = dense_edge_dets(rng, grid, dets)
    # the reference calls the model only for tiles that survive preprocessing + the row check, so
    # replay by tile id instead of call order
    class TileReplay(ReplayModel):
        def __init__(self, dets):
            ReplayModel.__init__(self, dets)
            self.cur = -1
    model = TileReplay(dets)
    dp = rpp.DataPreprocessor(PIPELINES["zscale_minmax"]())
    cfg = base_config(path, preprocess_fcn=dp, img_size=tsize, score_thr=0.7, iou_thr=0.5,
                      split_image_in_tiles=True, tile_xsize=tsize, tile_ysize=tsize,
                      tile_xstep=step, tile_ystep=step, devices=["cpu"])
    cfg["mpi"] = FakeMPI(1, 0)
    sf = SFinder(model, cfg)
    called = []
    # hook find_sources to set the replay cursor to the tile id
    from caesar_yolo.inference import TileTask
    orig_fs = TileTask.find_sources

    def fs(self):
        model.k = self.tid
        n0 = len(model.seen)
        r = orig_fs(self)
        called.append([int(self.tid), int(r), len(model.seen) - n0])
        return r
    TileTask.find_sources = fs
    snap = {}
    orig_merge = SFinder.merge_edge_sources

    def merge(self):
        snap["tile_sources"] = copy.deepcopy(self.tile_sources)
        return orig_merge(self)
    SFinder.merge_edge_sources = merge
    try:
        with quiet():
            assert sf.run_parallel() == 0
    finally:
        TileTask.find_sources = orig_fs
        SFinder.merge_edge_sources = orig_merge
    with open(os.path.join(scratch, "catalog_mosaic_%s.json" % tag)) as fp:
        cat_text = fp.read()

    def clean(o):
        if isinstance(o, dict):
            return {k: clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (np.bool_,)):
            return bool(o)
        return o
    fix = {"nx": nx, "ny": ny, "tile": tsize, "step": step, "grid": [list(map(int, t)) for t in grid],
           "dets": [[d[0].tolist(), d[1].tolist(), d[2].tolist()] for d in dets],
           "calls": called, "tile_sources_before_merge": clean(snap["tile_sources"]),
           "catalog_text": cat_text, "image_id": "mosaic_%s" % tag,
           "config": {"score_thr": 0.7, "iou_thr": 0.5, "soft": 0.3, "hard": 0.8, "img_size": tsize}}
    with open(os.path.join(outdir, "catalog_tiled_%s.json" % tag), "w") as fp:
        json.dump(fix, fp)
    if save_img:
        np.savez_compressed(os.path.join(outdir, "mosaic_%s.npz" % tag), img=img)
    nsrc = len(json.loads(cat_text)["sources"])
    print("catalog_tiled_%s.json: %d tiles, %d skipped, %d final sources, %d merged" % (
        tag, T, sum(1 for c in called if c[1] < 0), nsrc,
        sum(1 for s in json.loads(cat_text)["sources"] if s["merged"])))


def gen_wcs(outdir):
    """World coordinates of astropy.wcs.WCS(header) -- what the reference builds at inference.py:473 / utils.py:236, :411 -- for the
    header kinds its inputs carry (radio mosaics: SIN / TAN equatorial, CAR galactic; CD / PC / CROTA2 conventions; no WCS at all
    like test/galaxy0001.fits): pixel -> world and back at seeded pixel positions, both origins."""
    import warnings
    warnings.filterwarnings("ignore")
    from astropy.wcs import WCS as AW
    headers = {
        "tan": dict(CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CRPIX1=500.5, CRPIX2=400.5, CRVAL1=266.4, CRVAL2=-28.9, CDELT1=-0.000555, CDELT2=0.000555),
        "sin": dict(CTYPE1="RA---SIN", CTYPE2="DEC--SIN", CRPIX1=321.0, CRPIX2=123.0, CRVAL1=12.5, CRVAL2=-45.0, CDELT1=-0.001, CDELT2=0.001),
        "car_gal": dict(CTYPE1="GLON-CAR", CTYPE2="GLAT-CAR", CRPIX1=2000.0, CRPIX2=300.0, CRVAL1=5.5, CRVAL2=0.0, CDELT1=-0.0016667, CDELT2=0.0016667),
        "car_off": dict(CTYPE1="GLON-CAR", CTYPE2="GLAT-CAR", CRPIX1=100.0, CRPIX2=50.0, CRVAL1=340.0, CRVAL2=1.5, CDELT1=-0.002, CDELT2=0.002),
        "tan_cd": dict(CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CRPIX1=10.0, CRPIX2=20.0, CRVAL1=83.6, CRVAL2=22.0, CD1_1=-0.0004, CD1_2=0.0001, CD2_1=0.0001, CD2_2=0.0004),
        "tan_pc": dict(CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CRPIX1=256.0, CRPIX2=256.0, CRVAL1=201.3, CRVAL2=-43.0, CDELT1=-0.0008, CDELT2=0.0008,
                       PC1_1=0.9, PC1_2=-0.43589, PC2_1=0.43589, PC2_2=0.9),
        "sin_rot": dict(CTYPE1="RA---SIN", CTYPE2="DEC--SIN", CRPIX1=512.0, CRPIX2=512.0, CRVAL1=0.2, CRVAL2=70.0, CDELT1=-0.0011, CDELT2=0.0011, CROTA2=15.0),
        "tan_lonpole": dict(CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CRPIX1=50.0, CRPIX2=60.0, CRVAL1=150.0, CRVAL2=2.2, CDELT1=-0.0005, CDELT2=0.0005, LONPOLE=170.0),
        "sfl": dict(CTYPE1="GLON-SFL", CTYPE2="GLAT-SFL", CRPIX1=300.0, CRPIX2=200.0, CRVAL1=30.0, CRVAL2=0.0, CDELT1=-0.003, CDELT2=0.003),
        "arc": dict(CTYPE1="RA---ARC", CTYPE2="DEC--ARC", CRPIX1=30.0, CRPIX2=20.0, CRVAL1=45.0, CRVAL2=-10.0, CDELT1=-0.01, CDELT2=0.01),
        "linear": dict(CRPIX1=1.0, CRPIX2=1.0, CRVAL1=0.0, CRVAL2=0.0, CDELT1=1.0, CDELT2=1.0),
        "galaxy0001_like": dict(BMAJ=0.00416, BMIN=0.00416, BPA=0.0),           # no WCS keywords at all (test/galaxy0001.fits)
    }
    rng = np.random.default_rng(404)
    out = {}
    for name, kw in headers.items():
        h = _afits.Header()
        h["NAXIS"] = 2; h["NAXIS1"] = 1000; h["NAXIS2"] = 800
        for k, v in kw.items():
            h[k] = v
        w = AW(h)
        x, y = rng.uniform(-50, 1200, 24), rng.uniform(-50, 900, 24)
        rec = {"header": {k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in kw.items()}, "x": x.tolist(), "y": y.tolist()}
        for origin in (0, 1):
            a, d = w.all_pix2world(x, y, origin)
            px, py = w.wcs_world2pix(a, d, origin)
            rec["world_%d" % origin] = [np.asarray(a).tolist(), np.asarray(d).tolist()]
            rec["pix_back_%d" % origin] = [np.asarray(px).tolist(), np.asarray(py).tolist()]
        rec["pixel_scale_matrix"] = np.asarray(w.pixel_scale_matrix).tolist()
        out[name] = rec
    with open(os.path.join(outdir, "wcs.json"), "w") as fp:
        json.dump(out, fp)
    print("wcs.json: %d headers (astropy %s)" % (len(out), __import__("astropy").__version__))


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else "/root/repo/tests/golden"
    scratch = "/tmp/caesar_golden_scratch"
    os.makedirs(scratch, exist_ok=True)
    os.makedirs(outdir, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[2] == "round4":      # additions of round 4 only: world coordinates of astropy.wcs.WCS(header)
        gen_wcs(outdir)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "round2":      # additions of round 2 only (the other fixtures stay byte-identical)
        gen_preproc_chid(outdir)
        run_tiled(outdir, scratch, "d", 1500, 1100, 256, 0.5, 34, nper=2, dense=True, save_img=False)
        return
    gen_preproc(outdir)
    gen_tiles(outdir, scratch)
    gen_process_detections(outdir)
    run_serial(outdir, scratch)
    run_tiled(outdir, scratch, "a", 1300, 900, 512, 0.8, 31)
    run_tiled(outdir, scratch, "b", 1024, 1024, 256, 1.0, 32, nper=8)
    run_tiled(outdir, scratch, "c", 900, 700, 256, 0.5, 33, nper=6)
    gen_preproc_chid(outdir)
    run_tiled(outdir, scratch, "d", 1500, 1100, 256, 0.5, 34, nper=2, dense=True, save_img=False)
    gen_wcs(outdir)


if __name__ == "__main__":
    main()
