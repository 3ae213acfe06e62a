"""Host-side pieces that need no GPU: FITS ingest, the CLI's stage lowering, the tile partition."""
import os
import numpy as np
import pytest
from caesar_yolo_amd import utils, lib as L
from caesar_yolo_amd import preprocessing as PP
from caesar_yolo_amd.inference import partition_tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fits_reader_on_reference_fixture(golden_dir):
    data, hdr = utils.read_fits_image(os.path.join(golden_dir, "galaxy0001.fits"))
    ref = np.load(os.path.join(golden_dir, "preproc.npz"))["in/galaxy"]
    assert data.shape == (132, 132) and data.dtype == np.dtype(">f4")
    assert np.array_equal(np.asarray(data, np.float32), ref)
    assert hdr["NAXIS1"] == 132 and hdr["BITPIX"] == -32 and abs(hdr["BMAJ"] - 0.002611826449586) < 1e-15
    assert utils.image_id_of("/a/b/galaxy0001.fits") == "galaxy0001"


def test_fits_roundtrip_and_degenerate_axes(tmp_path):
    img = np.random.default_rng(0).normal(size=(37, 53)).astype(np.float32)
    img[3, 4] = np.nan
    p = str(tmp_path / "a.fits")
    utils.write_fits_image(p, img, [("BMAJ", 0.01), ("BUNIT", "Jy/beam")])
    data, hdr = utils.read_fits_image(p)
    assert np.array_equal(np.asarray(data, np.float32), img, equal_nan=True) and hdr["BUNIT"] == "Jy/beam"
    # 4-D cube with two degenerate leading axes: plane [0,0] is the image (utils.py:207-210 of the reference)
    raw = open(p, "rb").read()
    hdr_txt = raw[:2880].decode()
    hdr_txt = hdr_txt.replace("NAXIS   =                    2", "NAXIS   =                    4")
    cards = [hdr_txt[i:i + 80] for i in range(0, 2880, 80)]
    end = next(i for i, c in enumerate(cards) if c.startswith("END"))
    cards[end:end] = [("%-8s= %20d" % ("NAXIS3", 1)).ljust(80), ("%-8s= %20d" % ("NAXIS4", 1)).ljust(80)]
    q = str(tmp_path / "b.fits")
    open(q, "wb").write("".join(cards)[:2880].encode() + raw[2880:])
    data4, hdr4 = utils.read_fits_image(q)
    assert hdr4["NAXIS"] == 4 and np.array_equal(np.asarray(data4, np.float32), img, equal_nan=True)
    assert utils.read_fits_image(str(tmp_path / "missing.fits")) is None


def test_stage_lowering_matches_cli_order():
    dp = PP.DataPreprocessor([PP.BkgSubtractor(sigma=3), PP.SigmaClipShifter(sigma=1), PP.SigmaClipper(10, 10),
                              PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)])
    cfg = dp.program()
    assert cfg.nprog == 1 and cfg.prog[0].n == 5
    assert [cfg.prog[0].st[i].op for i in range(5)] == [L.OP_BKG, L.OP_SHIFT, L.OP_CLIP, L.OP_ZSCALE, L.OP_MINMAX]
    c3 = PP.DataPreprocessor([PP.ChanResizer(3), PP.Chan3Trasformer(0, 10, 10, 0.25), PP.MinMaxNormalizer(0, 255)]).program()
    assert c3.nprog == 3 and [c3.prog[i].n for i in range(3)] == [3, 3, 2]
    assert [c3.prog[2].st[i].op for i in range(2)] == [L.OP_HISTEQ, L.OP_MINMAX]
    assert PP.DataPreprocessor([]).program().nprog == 0
    # --bkg_chid / --clip_chid: the stage goes into that channel's program only (preprocessing.py:653, :712, :766)
    ch = PP.DataPreprocessor([PP.BkgSubtractor(sigma=3, chid=2), PP.SigmaClipper(10, 10, chid=1), PP.MinMaxNormalizer(0, 255)]).program()
    assert ch.nprog == 3 and [ch.prog[i].n for i in range(3)] == [1, 2, 2]
    assert [ch.prog[1].st[i].op for i in range(2)] == [L.OP_CLIP, L.OP_MINMAX]
    assert [ch.prog[2].st[i].op for i in range(2)] == [L.OP_BKG, L.OP_MINMAX]
    with pytest.raises(ValueError):
        PP.SigmaClipper(10, 10, chid=3)
    with pytest.raises(RuntimeError):
        dp(np.zeros((4, 4, 3)))                   # there is no CPU preprocessing path in the product


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_is_balanced_and_complete(world):
    """Every tile exactly once; contiguous runs of few shape classes; ESTIMATED TIME (pixels + a fixed overhead per batch,
    inference._rank_cost) within a few tiles of each other, so the rank that inherits the small ragged classes gets
    fewer full tiles instead of finishing last."""
    from caesar_yolo_amd import inference as I
    grid = utils.generate_tiles(0, 16383, 0, 16383, 512, 512, 0.8, 0.8)
    parts = partition_tiles(grid, 512, world, 256)
    assert sorted(t for p in parts for _, tids in p for t in tids) == list(range(len(grid)))
    area = {}
    for p in parts:
        for (th, tw), tids in p:
            lb = L.letterbox(th, tw, 512)
            area[(th, tw)] = lb.H * lb.W
    cost = {k: v / float(max(area.values())) for k, v in area.items()}
    est = [I._rank_cost([(shp, len(t)) for shp, t in p], cost, 256) for p in parts]
    # the makespan is what is balanced; the rank filled last may stay under it (one more tile can open a new batch)
    assert max(est) - sum(est) / len(est) <= 10.0, est      # tile-equivalents
    assert all(len(p) <= 4 for p in parts)
    if world == 8:                                           # the ragged classes cost their rank three extra launch sequences
        full = [sum(len(t) for shp, t in p if shp == (512, 512)) for p in parts]
        assert full[-1] < min(full[:-1]) - 60
    assert partition_tiles(grid, 512, world, 256) == parts   # deterministic: every rank derives the same assignment


def test_ds9_region_writer(tmp_path):
    """DS9 box regions from catalog objects (caesar_yolo/inference.py:1214-1263): centre/size arithmetic, 1-based image
    coordinates, tags and class colours; an empty list writes nothing."""
    from caesar_yolo_amd import utils
    objs = [dict(name="S1", x1=10, x2=30, y1=5, y2=15, class_name="compact", edge=False, merged=False),
            dict(name="S2_t3", x1=0, x2=7, y1=100, y2=103, class_name="extended", edge=True, merged=True)]
    out = tmp_path / "r.reg"
    assert utils.write_ds9_regions(str(out), objs) == 2
    lines = out.read_text().splitlines()
    assert lines[0].startswith("# Region file format: DS9") and lines[1] == "image"
    assert lines[2] == "box(21.0000,11.0000,20.0000,10.0000,0.00000000) # text={S1} tag={compact} color=blue"
    assert lines[3] == "box(4.5000,102.5000,7.0000,3.0000,0.00000000) # text={S2_t3} tag={extended} tag={BORDER} tag={MERGED} color=green"
    assert utils.ds9_region_lines(objs[1:], merged_tag=False)[0].count("tag=") == 2      # Analyzer regions carry no MERGED tag
    assert utils.write_ds9_regions(str(tmp_path / "e.reg"), []) == 0 and not (tmp_path / "e.reg").exists()


def test_beam_metadata_from_header(golden_dir, tmp_path):
    """SFinder's beam/pixel metadata (caesar_yolo/inference.py:430-468): the formula on a header with all five keywords, and
    the all-keywords-or-nothing rule on the reference's own FITS fixture (it has BMAJ/BMIN/BPA but no CDELT1/2 -> 0)."""
    from caesar_yolo_amd.inference import SFinder
    sf = SFinder.__new__(SFinder)
    _, sf.header = utils.read_fits_image(os.path.join(golden_dir, "galaxy0001.fits"))
    sf._beam_info()
    assert sf.beamArea == 0 and sf.pa == sf.header["BPA"]
    img = np.zeros((8, 8), np.float32)
    cards = [("CDELT1", -2.777777777778e-4), ("CDELT2", 2.777777777778e-4), ("BMAJ", 0.002611826449586),
             ("BMIN", 0.002142504259875), ("BPA", 84.46066805677)]
    utils.write_fits_image(str(tmp_path / "beam.fits"), img, cards)
    _, sf.header = utils.read_fits_image(str(tmp_path / "beam.fits"))
    sf._beam_info()
    h = dict(cards)
    expect = np.pi * h["BMAJ"] * h["BMIN"] / (4 * np.log(2)) / abs(h["CDELT1"] * h["CDELT2"])
    assert abs(sf.beamArea - expect) <= 1e-12 * expect and 80.0 < sf.beamArea < 90.0


def test_generate_tiles_matches_reference_golden(golden_dir):
    """The PRODUCT's utils.generate_tiles (not the oracle's copy) against the grids the imported reference produced
    (tests/golden/tiles.json, oracle/gen_golden.py: caesar_yolo/utils.py:622-697), incl. the three BASELINE grids."""
    import json
    from caesar_yolo_amd import utils
    fx = json.load(open(os.path.join(golden_dir, "tiles.json")))
    assert len(fx) >= 9
    for name, case in fx.items():
        got = utils.generate_tiles(*case["args"])
        want = case["tiles"]
        if want is None:
            assert got is None, name
        else:
            assert got is not None and [list(t) for t in got] == [list(t) for t in want], name


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_mosaic_source_uploads_only_the_ranks_regions(world):
    """each rank holds only the rows its tiles touch.  With a MosaicSource the engine uploads the bounding box of
    each shape-class segment of ITS tiles; at world 8 that is about 1/8 of the S16k mosaic (+ the overlap band), tile origins
    are rebased to the region, and together the ranks still cover every tile."""
    import torch
    from caesar_yolo_amd.inference import TileEngine, MosaicSource

    class FakeDet(object):
        max_batch = 256
        tdev = torch.device("cpu")

        def __init__(self):
            self.calls = []

        def mosaic_to_device(self, arr, big_endian=None, file_rows=None):
            return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))

        def detect_tiles(self, img, xy, th, tw, imgsz, cfg, conf, iou, soft, hard, out=None, flush=True):
            for x, y in xy:
                assert 0 <= x and x + tw <= img.shape[1] and 0 <= y and y + th <= img.shape[0]
                self.calls.append((float(img[y, x]), th, tw))          # the pixel at the tile origin identifies the tile
            return out

    n = 4096
    host = (np.arange(n, dtype=np.float32)[:, None] * n + np.arange(n, dtype=np.float32)[None, :])   # value = y * n + x (exact in fp32)
    grid = utils.generate_tiles(0, n - 1, 0, n - 1, 512, 512, 0.8, 0.8)
    seen, total = [], 0
    for rank in range(world):
        det, src = FakeDet(), MosaicSource(host)
        eng = TileEngine(det, src, grid, None, 512, 0.7, 0.5, 0.3, 0.8, rank, world, batch=256)
        eng.run_local()
        for (_, tids) in eng.my:
            for t in tids:
                seen.append(t)
        want = sorted((float(grid[t][2] * n + grid[t][0]), grid[t][3] - grid[t][2], grid[t][1] - grid[t][0]) for _, tids in eng.my for t in tids)
        assert sorted(det.calls) == want                                # every tile was cropped at its own origin
        total += src.bytes_uploaded
        if world == 8:
            assert src.bytes_uploaded <= 0.32 * host.nbytes            # on this small mosaic a two-row band is already 30 %
    assert sorted(seen) == list(range(len(grid)))
    if world == 1:
        assert total <= 1.25 * host.nbytes                              # full tiles + the two ragged strips + the corner


def test_rank_regions_of_the_16k_mosaic_are_an_eighth_plus_halo():
    """The same rule on the benchmark grid, as arithmetic on the partition (no 1 GiB array): at 8 ranks every rank's regions
    are at most 18 % of the 16384^2 mosaic (1/8 = 12.5 % + the 102-px overlap bands; the last rank also holds the ragged strips)."""
    from caesar_yolo_amd.inference import partition_tiles
    n = 16384
    grid = utils.generate_tiles(0, n - 1, 0, n - 1, 512, 512, 0.8, 0.8)
    for part in partition_tiles(grid, 512, 8, 256):
        px = 0
        for _, tids in part:
            g = [grid[t] for t in tids]
            px += (max(t[1] for t in g) - min(t[0] for t in g)) * (max(t[3] for t in g) - min(t[2] for t in g))
        assert px <= 0.18 * n * n


def test_mosaic_source_widens_nearly_full_width_bands_to_whole_file_rows(tmp_path):
    """Host logic of the ingest path (no GPU): a band that covers >= 90 % of the image width of a memory-mapped FITS payload is
    widened to whole rows and handed to the upload together with (file, byte offset of its first row) for the pread() path;
    a narrower region stays a view of the map with no file hint; the returned origin is the region's, for rebasing tiles."""
    import numpy as np
    from caesar_yolo_amd import utils
    from caesar_yolo_amd.inference import MosaicSource
    img = np.arange(200 * 320, dtype=np.float32).reshape(200, 320)
    path = str(tmp_path / "w.fits")
    utils.write_fits_image(path, img, {})
    data, _ = utils.read_fits_image(path)
    assert isinstance(data, np.memmap)
    calls = []

    class FakeDet(object):
        def mosaic_to_device(self, arr, big_endian=None, file_rows=None):
            calls.append((arr.shape, file_rows, bool(big_endian)))
            return np.asarray(arr).astype("<f4")

    src = MosaicSource(data)
    t, ox, oy = src.region(FakeDet(), 0, 300, 10, 150)          # 300 of 320 columns
    assert (ox, oy) == (0, 10) and t.shape == (140, 320)
    shape, file_rows, be = calls[-1]
    assert shape == (140, 320) and be and file_rows == (path, int(data.offset) + 10 * 320 * 4)
    assert np.array_equal(t, img[10:150])
    t2, ox2, oy2 = src.region(FakeDet(), 200, 320, 0, 200)       # 120 of 320 columns: a plain view, own origin
    assert (ox2, oy2) == (200, 0) and calls[-1][0] == (200, 120) and calls[-1][1] is None
    assert np.array_equal(t2, img[:, 200:320])
    assert src.bytes_uploaded == (140 * 320 + 200 * 120) * 4
