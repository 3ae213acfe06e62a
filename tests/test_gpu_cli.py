"""scripts/run.py end to end on the GPU (the reference's CLI surface, test/run_inference.sh / run_inference_parallel.sh
shaped invocations): serial frame and tiled mosaic, catalogs written where the reference writes them."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
from gpu_common import ROOT

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run.py")] + args, cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    return r.returncode, r.stdout.decode(errors="replace")


def test_cli_serial_and_tiled(tmp_path):
    from caesar_yolo_amd import utils
    # serial: test/run_inference.sh with BASELINE config-1 thresholds
    rc, out = _run(["--image=" + os.path.join(ROOT, "tests/golden/galaxy0001.fits"), "--weights=seeded:l:5",
                    "--preprocessing", "--zscale_stretch", "--zscale_contrasts=0.25,0.25,0.25", "--normalize_minmax",
                    "--norm_min=0", "--norm_max=255", "--imgsize=640", "--scoreThr=0.7", "--iouThr=0.5", "--devices=0"],
                   str(tmp_path))
    assert rc == 0, out[-2000:]
    cat = json.load(open(tmp_path / "out_galaxy0001.json"))
    assert cat["image_id"] == "galaxy0001" and len(cat["objs"]) > 0
    assert set(cat["objs"][0]) == {"name", "x1", "x2", "y1", "y2", "class_id", "class_name", "score", "edge"}
    # tiled: test/run_inference_parallel.sh shape (256x256 tiles, step 1) on a synthetic mosaic
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_b.npz"))["img"]
    utils.write_fits_image(str(tmp_path / "mosaic_b.fits"), img)
    rc, out = _run(["--image=" + str(tmp_path / "mosaic_b.fits"), "--weights=seeded:l:5", "--preprocessing", "--zscale_stretch",
                    "--normalize_minmax", "--norm_max=255", "--imgsize=256", "--split_img_in_tiles", "--tile_xsize=256",
                    "--tile_ysize=256", "--tile_xstep=1", "--tile_ystep=1", "--devices=0", "--tile_batch=16"], str(tmp_path))
    assert rc == 0, out[-2000:]
    src = json.load(open(tmp_path / "catalog_mosaic_b.json"))["sources"]
    assert len(src) > 0 and [s["name"] for s in src] == ["S%d" % (i + 1) for i in range(len(src))]
    assert all(0 <= s["x1"] <= s["x2"] <= 1024 and 0 <= s["y1"] <= s["y2"] <= 1024 for s in src)
    # argument validation mirrors the reference: a missing image is an error exit, not a crash
    rc, _ = _run(["--image=" + str(tmp_path / "nope.fits"), "--weights=seeded:l:5"], str(tmp_path))
    assert rc == 1


def test_cli_reference_defaults_and_tile_outputs(tmp_path):
    """The reference's own defaults and optional outputs: --devices=cpu (scripts/run.py:130; here: GPU with a warning),
    --save_tile_catalog / --save_tile_region / --save_tile_img (inference.py:220-229, :330-350), --bkg_chid, and the
    max_ntasks_per_worker guard (inference.py:1151-1160) when it is given explicitly."""
    from caesar_yolo_amd import utils
    img = np.load(os.path.join(ROOT, "tests/golden/mosaic_b.npz"))["img"]
    utils.write_fits_image(str(tmp_path / "mosaic_b.fits"), img)
    base = ["--image=" + str(tmp_path / "mosaic_b.fits"), "--weights=seeded:l:5", "--preprocessing", "--subtract_bkg", "--bkg_chid=1",
            "--zscale_stretch", "--normalize_minmax", "--norm_max=255", "--imgsize=256", "--split_img_in_tiles", "--tile_xsize=256",
            "--tile_ysize=256", "--tile_xstep=1", "--tile_ystep=1", "--tile_batch=16", "--scoreThr=0.3", "--precision=fp16"]
    rc, out = _run(base + ["--save_tile_catalog", "--save_tile_region", "--save_tile_img"], str(tmp_path))     # no --devices: the default "cpu"
    assert rc == 0, out[-2000:]
    assert "no CPU path" in out
    src = json.load(open(tmp_path / "catalog_mosaic_b.json"))["sources"]
    assert len(src) > 0
    tiles = sorted(f for f in os.listdir(tmp_path) if f.startswith("catalog_mosaic_b_tid") and f.endswith(".json"))
    assert 1 <= len(tiles) <= 16
    n_objs = 0
    for f in tiles:
        t = json.load(open(tmp_path / f))
        tid = int(f[len("catalog_mosaic_b_tid"):-5])
        assert t["image_id"] == "mosaic_b"
        for o in t["objs"]:
            assert o["name"].endswith("_t%d" % tid)
        n_objs += len(t["objs"])
        assert os.path.exists(tmp_path / ("timg_mosaic_b_tid%d.fits" % tid))
        assert os.path.exists(tmp_path / ("catalog_mosaic_b_tid%d.reg" % tid)) == (len(t["objs"]) > 0)
    assert n_objs >= len(src)                                   # the cross-tile merge can only reduce the count
    tile0, _ = utils.read_fits_image(str(tmp_path / ("timg_mosaic_b_tid%d.fits" % int(tiles[0][len("catalog_mosaic_b_tid"):-5]))))
    assert tile0.shape == (256, 256)
    rc, out = _run(base + ["--devices=0", "--max_ntasks_per_worker=4"], str(tmp_path))                           # 16 tiles on one rank > 4
    assert rc == 1 and "Too many tasks per worker" in out
