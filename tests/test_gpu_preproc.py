"""Preprocessing parity through the C-ABI: cy_preproc (statistics + apply + letterbox/pack, csrc/cy_preproc.hip) against
the numpy oracle, whose outputs are themselves pinned to vectors captured from the imported reference.
Solved statistics (zscale limits, sigma-clip bounds, ...) agree to 1e-11 relative; the packed fp32 network input agrees
to 1 fp32 ulp of the /255 value (2e-7 absolute); the zero mask is identical."""
import os
import numpy as np
import pytest
import torch
from gpu_common import detector, ROOT
from caesar_yolo_amd import preprocessing as PP
from caesar_yolo_amd import lib as L

pytestmark = pytest.mark.gpu

PIPES = {
    "zscale_minmax": lambda M: [M.ZScaleTransformer(contrasts=[0.25] * 3), M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "full": lambda M: [M.BkgSubtractor(sigma=3), M.SigmaClipShifter(sigma=1), M.SigmaClipper(sigma_low=10, sigma_up=10),
                       M.ZScaleTransformer(contrasts=[0.25] * 3), M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "chan3_minmax": lambda M: [M.ChanResizer(nchans=3), M.Chan3Trasformer(sigma_clip_baseline=0, sigma_clip_low=10,
                                                                          sigma_clip_up=10, zscale_contrast=0.25),
                               M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "bkg_box": lambda M: [M.BkgSubtractor(sigma=3, use_mask_box=True, mask_fract=0.7)],
    "clip_1_3": lambda M: [M.SigmaClipper(sigma_low=1, sigma_up=3)],
    "minmax": lambda M: [M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "zscale_c40": lambda M: [M.ZScaleTransformer(contrasts=[0.4] * 3)],
}
ORACLE = {
    "zscale_minmax": [("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))],
    "full": [("bkg", dict(sigma=3)), ("shift", dict(sigma=1)), ("clip", dict(sigma_low=10, sigma_up=10)),
             ("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))],
    "chan3_minmax": [("chanresize", dict(nchans=3)), ("chan3", dict(sigma_clip_baseline=0, sigma_clip_low=10,
                                                                   sigma_clip_up=10, zscale_contrast=0.25)),
                     ("minmax", dict(norm_min=0, norm_max=255))],
    "bkg_box": [("bkg", dict(sigma=3, use_mask_box=True, mask_fract=0.7))],
    "clip_1_3": [("clip", dict(sigma_low=1, sigma_up=3))],
    "minmax": [("minmax", dict(norm_min=0, norm_max=255))],
    "zscale_c40": [("zscale", dict(contrasts=[0.4] * 3))],
}


def _mosaic(det, tiles):
    """Lay the test tiles side by side in one resident mosaic (with a border so origins are non-trivial)."""
    h = max(t.shape[0] for t in tiles) + 7
    w = sum(t.shape[1] for t in tiles) + 5 * (len(tiles) + 1)
    m = np.full((h, w), np.nan, np.float32)          # NaN background: exercises non-finite -> 0 on ingest
    xy, x = [], 5
    for t in tiles:
        m[3:3 + t.shape[0], x:x + t.shape[1]] = t
        xy.append((x, 3))
        x += t.shape[1] + 5
    return det.mosaic_to_device(m.astype(">f4")), xy      # big-endian like a FITS payload


@pytest.mark.parametrize("pname", sorted(PIPES))
@pytest.mark.parametrize("iname", ["galaxy", "syn192", "rag", "dense"])
def test_preproc_matches_oracle(pname, iname):
    from oracle import preprocessing_ref as P
    from oracle import yolov8_ref as Y
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/" + iname]
    th, tw = img.shape
    imgsz = 256
    mosaic, xy = _mosaic(det, [img, img[::-1].copy()])
    cfg = PP.DataPreprocessor(PIPES[pname](PP)).program()
    netin, status, lb = det.preproc(mosaic, xy, th, tw, imgsz, cfg)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 0]
    for b, src in enumerate([img, img[::-1].copy()]):
        ref_img = P.build_pipeline(ORACLE[pname])(P.to_cube(src))
        if b == 0 and not pname.startswith("chan3"):
            # the oracle itself is pinned to the reference's output for this case
            np.testing.assert_allclose(ref_img[:, :, 0], g["out/%s/%s" % (iname, pname)], rtol=1e-11, atol=1e-13)
        x, hw = Y.preprocess(ref_img, imgsz)                       # [1,3,H,W] fp32, flipped, /255
        assert hw == (lb.H, lb.W)
        got = netin[b].float().cpu().numpy()                       # [H,W,4]
        ref = x[0].permute(1, 2, 0).numpy()
        assert np.all(got[..., 3] == 0)
        if lb.new_h == th and lb.new_w == tw:
            assert np.array_equal(got[..., :3] == 0, ref == 0)
            np.testing.assert_allclose(got[..., :3], ref, rtol=0, atol=2e-7)
        else:
            np.testing.assert_allclose(got[..., :3], ref, rtol=0, atol=5e-7)


def test_solved_statistics_match_reference_stats():
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    for iname in ["galaxy", "syn192", "rag", "dense", "big512"]:
        img = g["in/" + iname]
        th, tw = img.shape
        mosaic, xy = _mosaic(det, [img])
        cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3)]).program()
        det.preproc(mosaic, xy, th, tw, 640, cfg)
        p = det.preproc_params(1)[0, 0]
        np.testing.assert_allclose(p[0, :2], g["stats/%s/zscale_0.25" % iname], rtol=1e-11)
        cfg = PP.DataPreprocessor([PP.SigmaClipper(10, 10)]).program()
        det.preproc(mosaic, xy, th, tw, 640, cfg)
        p = det.preproc_params(1)[0, 0]
        np.testing.assert_allclose(p[0, :2], g["stats/%s/sigclip_bounds_10_10" % iname], rtol=1e-11)
        cfg = PP.DataPreprocessor([PP.BkgSubtractor(sigma=3)]).program()
        det.preproc(mosaic, xy, th, tw, 640, cfg)
        p = det.preproc_params(1)[0, 0]
        np.testing.assert_allclose(p[0, 0], g["stats/%s/sigstats_3" % iname][0], rtol=1e-10)


def test_tile_rejection_status():
    """All-zero tile -> MinMaxNormalizer returns None (status 1); constant first rows -> Analyzer's row check (status 2)."""
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/syn192"].copy()
    z = np.zeros_like(img)
    q = img.copy()
    q[1, :] = 0.0
    ok5 = img.copy()
    ok5[5, :] = 0.0
    mosaic, xy = _mosaic(det, [img, z, q, ok5])
    cfg = PP.DataPreprocessor(PIPES["zscale_minmax"](PP)).program()
    _, status, _ = det.preproc(mosaic, xy, 192, 192, 192, cfg)
    assert status.cpu().tolist() == [0, 1, 2, 0]


def test_big512_sample_fp32():
    from oracle import preprocessing_ref as P
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/big512"]
    mosaic, xy = _mosaic(det, [img])
    for pname in ("zscale_minmax", "chan3_minmax", "full"):
        cfg = PP.DataPreprocessor(PIPES[pname](PP)).program()
        netin, status, lb = det.preproc(mosaic, xy, 512, 512, 512, cfg)
        got = netin[0].float().cpu().numpy()[..., :3][..., ::-1].reshape(-1, 3)[::97]
        ref = (g["big512_sample/%s" % pname].astype(np.float32) / np.float32(255)).astype(np.float32)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)


CHID_PIPES = {
    "bkg_chid0": lambda M: [M.BkgSubtractor(sigma=3, chid=0)],
    "shiftclip_chid1_minmax": lambda M: [M.SigmaClipShifter(sigma=1, chid=1), M.SigmaClipper(sigma_low=10, sigma_up=10, chid=1),
                                         M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "bkg_chid2_zscale_minmax": lambda M: [M.BkgSubtractor(sigma=3, chid=2), M.ZScaleTransformer(contrasts=[0.25] * 3),
                                          M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "bkgbox_chid1_clip_chid0": lambda M: [M.BkgSubtractor(sigma=3, use_mask_box=True, mask_fract=0.7, chid=1),
                                          M.SigmaClipper(sigma_low=1, sigma_up=3, chid=0)],
}


@pytest.mark.parametrize("pname", sorted(CHID_PIPES))
@pytest.mark.parametrize("iname", ["galaxy", "syn192", "rag"])
def test_chid_stage_selection_matches_reference_golden(pname, iname):
    """--bkg_chid / --clip_chid (scripts/run.py:89, :98): compared DIRECTLY with the (H,W,3) outputs of the imported reference
    (tests/golden/preproc_chid.npz), through the model's LetterBox/flip//255."""
    from oracle import yolov8_ref as Y
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    gc = np.load(os.path.join(ROOT, "tests/golden/preproc_chid.npz"))
    img = g["in/" + iname]
    th, tw = img.shape
    imgsz = 192 if iname == "syn192" else 256          # syn192: no resize (exact zero mask); the others go through the resize
    mosaic, xy = _mosaic(det, [img])
    cfg = PP.DataPreprocessor(CHID_PIPES[pname](PP)).program()
    assert cfg.nprog == 3
    netin, status, lb = det.preproc(mosaic, xy, th, tw, imgsz, cfg)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0]
    ref_img = gc["out/%s/%s" % (iname, pname)]
    x, hw = Y.preprocess(ref_img, imgsz)
    assert hw == (lb.H, lb.W)
    got = netin[0].float().cpu().numpy()
    ref = x[0].permute(1, 2, 0).numpy()
    scale = 1.0
    if (lb.new_h, lb.new_w) == (th, tw):
        assert np.array_equal(got[..., :3] == 0, ref == 0)
        np.testing.assert_allclose(got[..., :3], ref, rtol=3e-7, atol=1e-30)       # un-normalised pipelines: values ~1e-7, so relative
    else:
        np.testing.assert_allclose(got[..., :3], ref, rtol=5e-7, atol=5e-7 * scale * float(np.abs(ref).max()))


GOLDEN_PIPES = {   # pipelines whose outputs the imported reference wrote into tests/golden/preproc.npz (oracle/gen_golden.py:PIPELINES)
    "bkg": lambda M: [M.BkgSubtractor(sigma=3)],
    "bkg_box": lambda M: [M.BkgSubtractor(sigma=3, use_mask_box=True, mask_fract=0.7)],
    "shift": lambda M: [M.SigmaClipShifter(sigma=1)],
    "clip": lambda M: [M.SigmaClipper(sigma_low=10, sigma_up=10)],
    "clip_1_3": lambda M: [M.SigmaClipper(sigma_low=1, sigma_up=3)],
    "zscale": lambda M: [M.ZScaleTransformer(contrasts=[0.25] * 3)],
    "zscale_c40": lambda M: [M.ZScaleTransformer(contrasts=[0.4] * 3)],
    "minmax": lambda M: [M.MinMaxNormalizer(norm_min=0, norm_max=255)],
    "zscale_minmax": PIPES["zscale_minmax"], "chan3_minmax": PIPES["chan3_minmax"], "full": PIPES["full"],
    "chan3_1_20": lambda M: [M.ChanResizer(nchans=3), M.Chan3Trasformer(sigma_clip_baseline=0, sigma_clip_low=1, sigma_clip_up=20,
                                                                        zscale_contrast=0.25)],
}


@pytest.mark.parametrize("pname", sorted(GOLDEN_PIPES))
@pytest.mark.parametrize("iname", ["galaxy", "syn192", "rag", "dense"])
def test_preprocessed_image_matches_reference_golden_fp64(pname, iname):
    """cy_preproc_planes: the device's preprocessed image in float64 against the REFERENCE's own output for the same stage list
    (tests/golden/preproc.npz, written by the imported caesar_yolo.preprocessing): identical zero mask, values to 1e-10
    relative (statistics are reductions in a different order; every per-pixel map is replayed operation for operation)."""
    det = detector("fp32", max_imgsz=640)
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/" + iname]
    th, tw = img.shape
    mosaic, xy = _mosaic(det, [img])
    cfg = PP.DataPreprocessor(GOLDEN_PIPES[pname](PP)).program()
    planes, status = det.preproc_planes(mosaic, xy, th, tw, cfg)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0]
    got = planes[0].cpu().numpy().transpose(1, 2, 0)               # (H, W, 3)
    ref = g["out/%s/%s" % (iname, pname)]
    if ref.ndim == 2:
        assert np.array_equal(got[:, :, 0], got[:, :, 1]) and np.array_equal(got[:, :, 0], got[:, :, 2])
        got = got[:, :, 0]
    assert np.array_equal(got == 0, ref == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-13)


def test_staged_upload_of_a_memory_map_slice(tmp_path, monkeypatch):
    """The chunked upload through pinned staging buffers (used for large images, here forced on a small one) gives the same
    resident mosaic as the plain upload: a non-contiguous slice of a big-endian memory map, several chunks per buffer, a
    last chunk that is shorter, NaN / inf -> 0 (utils.read_fits value semantics, caesar_yolo/utils.py:219, :394)."""
    from caesar_yolo_amd.model import HipDetector
    det = detector("fp32")
    rng = np.random.default_rng(5)
    full = rng.standard_normal((700, 900)).astype(">f4")
    full[3, 7] = np.nan
    full[650, 880] = np.inf
    path = tmp_path / "m.bin"
    full.tofile(path)
    mm = np.memmap(path, dtype=">f4", mode="r", shape=full.shape)
    view = mm[11:688, 5:893]                                   # 677 x 888: not contiguous
    plain = det.mosaic_to_device(np.ascontiguousarray(view), big_endian=True).cpu().numpy()
    monkeypatch.setattr(HipDetector, "STAGED_UPLOAD_MIN_BYTES", 1 << 10)
    monkeypatch.setattr(HipDetector, "STAGE_BYTES", 100 * 888 * 4)      # 100 rows per chunk -> 7 chunks, the last of 77 rows
    det._stage = None
    staged = det.mosaic_to_device(view, big_endian=True).cpu().numpy()
    det._stage = None
    want = np.nan_to_num(view.astype("<f4"), nan=0.0, posinf=0.0, neginf=0.0)
    assert staged.shape == want.shape
    assert np.array_equal(staged, want)
    assert np.array_equal(staged, plain)


def test_staged_upload_reads_whole_rows_with_pread(tmp_path, monkeypatch):
    """MosaicSource on a memory-mapped FITS payload: a (nearly) full-width band is widened to whole rows and read with pread()
    into the pinned buffers; the resident region equals the plain upload of the same rows and the origin is rebased."""
    from caesar_yolo_amd.model import HipDetector
    from caesar_yolo_amd.inference import MosaicSource
    from caesar_yolo_amd import utils
    det = detector("fp32")
    rng = np.random.default_rng(6)
    img = rng.standard_normal((300, 640)).astype(np.float32)
    img[5, 5] = np.nan
    path = str(tmp_path / "m.fits")
    utils.write_fits_image(path, img, {})
    data, _ = utils.read_fits_image(path)
    assert isinstance(data, np.memmap)
    monkeypatch.setattr(HipDetector, "STAGED_UPLOAD_MIN_BYTES", 1 << 10)
    monkeypatch.setattr(HipDetector, "STAGE_BYTES", 64 * 640 * 4)
    det._stage = None
    src = MosaicSource(data)
    t, ox, oy = src.region(det, 0, 600, 17, 290)               # 600 of 640 columns: widened to whole rows
    det._stage = None
    assert (ox, oy) == (0, 17) and tuple(t.shape) == (273, 640)
    want = np.nan_to_num(img[17:290], nan=0.0)
    assert np.array_equal(t.cpu().numpy(), want)
    t2, ox2, oy2 = src.region(det, 100, 300, 20, 200)           # inside the resident region: reused
    assert t2 is t and (ox2, oy2) == (0, 17)


@pytest.mark.parametrize("pname", ["zscale_minmax", "minmax", "clip_1_3", "bkg_box", "full"])
def test_lean_passes_agree_with_the_general_ones_other_pipelines(pname, monkeypatch):
    """The same switch on the other pipelines: MINMAX behind nothing / behind [ZSCALE] (raw_threshold), a sigma clip whose lower bound
    sits inside the noise (1 sigma), a masked box (general loop either way) and the five-stage chain (raw first stage only)."""
    from caesar_yolo_amd import synth
    det = detector("fp32", max_imgsz=640)
    mos = synth.make_mosaic(2048, seed=20260104)
    size = 512
    tiles = [np.ascontiguousarray(mos[y:y + size, x:x + size]) for (x, y) in [(0, 0), (900, 700)]]
    mosaic, xy = _mosaic(det, tiles)
    cfg = PP.DataPreprocessor(PIPES[pname](PP)).program()
    monkeypatch.setenv("CY_PRE_VARIANT", "0")
    base, st0 = det.preproc_planes(mosaic, xy, size, size, cfg)
    base, st0 = base.cpu().numpy(), st0.cpu().tolist()
    par0 = det.preproc_params(len(xy)).copy()
    for variant in (2, 4, 8, 64, 2 | 4 | 8 | 64):
        monkeypatch.setenv("CY_PRE_VARIANT", str(variant))
        got, st = det.preproc_planes(mosaic, xy, size, size, cfg)
        assert st.cpu().tolist() == st0
        np.testing.assert_allclose(got.cpu().numpy(), base, rtol=1e-12, atol=1e-12, err_msg="%s, variant %d" % (pname, variant))
        np.testing.assert_allclose(det.preproc_params(len(xy)), par0, rtol=1e-12, atol=0, err_msg="%s parameters, variant %d" % (pname, variant))


@pytest.mark.parametrize("size", [512, 640, 132])
def test_lean_passes_agree_with_the_general_ones(size, monkeypatch):
    """Round 4: the statistics passes over raw whole-group tiles have lean forms (moments_plain, minmax_plain, hist_plain, the first
    sigma-clip trip about the sample bracket's midpoint, the narrower median bracket of later clips, and, opt-in, the two sigma-clip programs of chan3 in one workgroup with a shared initial set = bit 4096).
    CY_PRE_VARIANT switches them off (on)
    one bit at a time (read per launch); the preprocessed float64 image must not care: the HISTEQ channel is bit-identical (same bins,
    same tables), the sigma-clip channels agree to 1e-12 (sums in another order, fma)."""
    from caesar_yolo_amd import synth
    det = detector("fp32", max_imgsz=640)
    mos = synth.make_mosaic(2048, seed=20260105)
    tiles = [np.ascontiguousarray(mos[y:y + size, x:x + size]) for (x, y) in [(0, 0), (700, 900), (1300, 200)]]
    mosaic, xy = _mosaic(det, tiles)
    cfg = PP.DataPreprocessor(PIPES["chan3_minmax"](PP)).program()
    monkeypatch.setenv("CY_PRE_VARIANT", "0")
    base, st0 = det.preproc_planes(mosaic, xy, size, size, cfg)
    base = base.cpu().numpy()
    par0 = det.preproc_params(len(xy)).copy()
    assert st0.cpu().tolist() == [0, 0, 0]
    for variant in (2, 4, 8, 64, 4096, 2 | 4 | 8 | 64 | 4096):
        monkeypatch.setenv("CY_PRE_VARIANT", str(variant))
        got, st = det.preproc_planes(mosaic, xy, size, size, cfg)
        got = got.cpu().numpy()
        assert st.cpu().tolist() == [0, 0, 0]
        np.testing.assert_array_equal(got[:, 2], base[:, 2], err_msg="HISTEQ channel, variant %d" % variant)
        np.testing.assert_allclose(got[:, :2], base[:, :2], rtol=1e-12, atol=1e-12, err_msg="sigma-clip channels, variant %d" % variant)
        np.testing.assert_allclose(det.preproc_params(len(xy)), par0, rtol=1e-12, atol=0, err_msg="solved parameters, variant %d" % variant)


def test_fast_division_is_the_division():
    """The pack kernel divides by per-tile constants with the tail of the float64 division's own instruction sequence (reciprocal and Newton
    steps hoisted: csrc/cy_preproc.hip fast_div).  Same bits as `a / b` on 2^22 pairs over 60 decades, zeros, denormals, infinities and
    exponents beyond the fast path's range (those take the division itself)."""
    import ctypes as C
    det = detector("fp32", max_imgsz=640)
    rng = np.random.default_rng(20260105)
    n = 1 << 22
    a = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-30, 30, n)
    b = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-30, 30, n)
    a[:4096] = 0.0
    a[4096:8192] *= 1e-290                                   # denormals and tiny values: the slow branch
    a[8192:12288] *= 1e250
    b[12288:16384] *= 1e200
    b[16384:16400] = [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 1.7e308] * 2
    a[16400:16416] = [np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-310, 1.7e308, -1.7e308] * 2
    # pixel-like operands: float32 values minus a float64 threshold, divided by a float64 range
    a[100000:600000] = rng.normal(0, 1e-3, 500000).astype(np.float32).astype(np.float64) - rng.normal(0, 1e-3)
    b[100000:600000] = rng.uniform(1e-4, 5e-2)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    fast, ref = torch.empty_like(ta), torch.empty_like(ta)
    rc = det.lib.cy_debug_fastdiv(C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(fast.data_ptr()), C.c_void_p(ref.data_ptr()), n)
    assert rc == 0
    f, r = fast.cpu().numpy().view(np.uint64), ref.cpu().numpy().view(np.uint64)
    nan = np.isnan(fast.cpu().numpy()) & np.isnan(ref.cpu().numpy())
    bad = np.nonzero((f != r) & ~nan)[0]
    assert bad.size == 0, "fast_div differs from a / b at %d of %d pairs, first: a=%r b=%r" % (bad.size, n, a[bad[0]], b[bad[0]])
    with np.errstate(all="ignore"):
        host = a / b
    ok = np.isfinite(host)
    assert np.array_equal(ref.cpu().numpy()[ok], host[ok])                # and the device division is IEEE (numpy's)


@pytest.mark.parametrize("pname", ["zscale_minmax", "chan3_minmax", "full"])
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_specialised_pack_is_the_general_pack(pname, prec, monkeypatch):
    """The pack kernel has forms specialised for the program shapes of the benchmarks (no per-stage switch; CY_PRE_VARIANT bit 8192 = the
    general form for every shape): the packed network input must be identical, bit for bit."""
    from caesar_yolo_amd import synth
    det = detector(prec, max_imgsz=640)
    mos = synth.make_mosaic(2048, seed=20260105)
    size = 640
    tiles = [np.ascontiguousarray(mos[y:y + size, x:x + size]) for (x, y) in [(0, 0), (1200, 900)]]
    mosaic, xy = _mosaic(det, tiles)
    cfg = PP.DataPreprocessor(PIPES[pname](PP)).program()
    monkeypatch.setenv("CY_PRE_VARIANT", "8192")
    ref, st0, _ = det.preproc(mosaic, xy, size, size, size, cfg)
    ref = ref.clone()
    monkeypatch.setenv("CY_PRE_VARIANT", "0")
    got, st, _ = det.preproc(mosaic, xy, size, size, size, cfg)
    assert st.cpu().tolist() == st0.cpu().tolist()
    assert torch.equal(got, ref)
