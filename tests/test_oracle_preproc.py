"""Pins oracle/preprocessing_ref.py against vectors captured from the imported reference
(tests/golden/preproc.npz, made by oracle/gen_golden.py)."""
import os
import numpy as np
import pytest
from oracle import preprocessing_ref as P

SPECS = {
    "bkg": [("bkg", dict(sigma=3))],
    "bkg_box": [("bkg", dict(sigma=3, use_mask_box=True, mask_fract=0.7))],
    "shift": [("shift", dict(sigma=1))],
    "clip": [("clip", dict(sigma_low=10, sigma_up=10))],
    "clip_1_3": [("clip", dict(sigma_low=1, sigma_up=3))],
    "zscale": [("zscale", dict(contrasts=[0.25] * 3))],
    "zscale_c40": [("zscale", dict(contrasts=[0.4] * 3))],
    "minmax": [("minmax", dict(norm_min=0, norm_max=255))],
    "zscale_minmax": [("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))],
    "chan3_minmax": [("chanresize", dict(nchans=3)),
                     ("chan3", dict(sigma_clip_baseline=0, sigma_clip_low=10, sigma_clip_up=10, zscale_contrast=0.25)),
                     ("minmax", dict(norm_min=0, norm_max=255))],
    "chan3_1_20": [("chanresize", dict(nchans=3)),
                   ("chan3", dict(sigma_clip_baseline=0, sigma_clip_low=1, sigma_clip_up=20, zscale_contrast=0.25))],
    "full": [("bkg", dict(sigma=3)), ("shift", dict(sigma=1)), ("clip", dict(sigma_low=10, sigma_up=10)),
             ("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))],
}
INPUTS = ["galaxy", "syn192", "rag", "dense"]


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "preproc.npz"))


def close(a, b, rtol=1e-11, atol=1e-13):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("iname", INPUTS)
@pytest.mark.parametrize("pname", sorted(SPECS))
def test_stage_outputs(g, iname, pname):
    img = g["in/" + iname]
    res = P.build_pipeline(SPECS[pname])(P.to_cube(img))
    ref = g["out/%s/%s" % (iname, pname)]
    if ref.ndim == 2:
        assert np.array_equal(res[:, :, 0], res[:, :, 1]) and np.array_equal(res[:, :, 0], res[:, :, 2])
        res = res[:, :, 0]
    # exact zero pattern (the cond mask) must match bit-for-bit
    assert np.array_equal(res == 0, ref == 0)
    close(res, ref)


@pytest.mark.parametrize("iname", INPUTS + ["big512"])
def test_zscale_limits(g, iname):
    d = g["in/" + iname].astype(np.float64)
    for c in ("0.25", "0.4"):
        key = "stats/%s/zscale_%s" % (iname, c)
        if key in g.files:
            close(np.array(P.zscale_limits(d, float(c))), g[key], rtol=1e-12)


def test_zscale_galaxy_known_values(g):
    # limits measured on galaxy0001.fits with the imported reference
    vmin, vmax = P.zscale_limits(g["in/galaxy"].astype(np.float64), 0.25)
    assert abs(vmin - (-2.0223498e-4)) < 1e-10 and abs(vmax - 4.116468830e-4) < 1e-10


@pytest.mark.parametrize("iname", INPUTS)
def test_sigma_clip_bounds_and_stats(g, iname):
    d = g["in/" + iname].astype(np.float64)
    nz = d[P.nonzero_finite(d)]
    for lo, up in ((10, 10), (1, 3), (0, 10), (1, 20), (3, 3), (1, 1)):
        _, a, b = P.sigma_clip_1d(nz, lo, up)
        close(np.array([a, b]), g["stats/%s/sigclip_bounds_%g_%g" % (iname, lo, up)], rtol=1e-12)
    for s in (3, 1):
        close(np.array(P.sigma_clipped_stats_1d(nz, s)), g["stats/%s/sigstats_%g" % (iname, s)], rtol=1e-11)


def test_big512_samples(g):
    img = g["in/big512"]
    for pname in ("zscale_minmax", "chan3_minmax", "full"):
        res = P.build_pipeline(SPECS[pname])(P.to_cube(img)).reshape(-1, 3)[::97]
        close(res, g["big512_sample/%s" % pname])


def test_all_zero_tile_is_rejected():
    z = np.zeros((16, 16), np.float32)
    assert P.build_pipeline(SPECS["zscale_minmax"])(P.to_cube(z)) is None


def test_row_check_quirk():
    img = np.random.default_rng(0).uniform(1, 2, (8, 8, 3))
    assert not P.rows_constant(img)
    img[1] = 0.0          # row 1 constant -> reference skips the tile (evaluation.py:171-176)
    assert P.rows_constant(img)
    img = np.random.default_rng(0).uniform(1, 2, (8, 8, 3))
    img[5] = 0.0          # only rows 0..2 are looked at
    assert not P.rows_constant(img)


# ---- per-channel stage selection (--bkg_chid / --clip_chid; caesar_yolo/preprocessing.py:594-601 + :653, :667-672 + :712,
# :726-732 + :766): tests/golden/preproc_chid.npz, captured from the imported reference by oracle/gen_golden.py
CHID_SPECS = {
    "bkg_chid0": [("bkg", dict(sigma=3, chid=0))],
    "shiftclip_chid1_minmax": [("shift", dict(sigma=1, chid=1)), ("clip", dict(sigma_low=10, sigma_up=10, chid=1)),
                               ("minmax", dict(norm_min=0, norm_max=255))],
    "bkg_chid2_zscale_minmax": [("bkg", dict(sigma=3, chid=2)), ("zscale", dict(contrasts=[0.25] * 3)),
                                ("minmax", dict(norm_min=0, norm_max=255))],
    "bkgbox_chid1_clip_chid0": [("bkg", dict(sigma=3, use_mask_box=True, mask_fract=0.7, chid=1)),
                                ("clip", dict(sigma_low=1, sigma_up=3, chid=0))],
}


@pytest.mark.parametrize("iname", ["galaxy", "syn192", "rag"])
@pytest.mark.parametrize("pname", sorted(CHID_SPECS))
def test_chid_stage_outputs(g, golden_dir, iname, pname):
    gc = np.load(os.path.join(golden_dir, "preproc_chid.npz"))
    res = P.build_pipeline(CHID_SPECS[pname])(P.to_cube(g["in/" + iname]))
    ref = gc["out/%s/%s" % (iname, pname)]
    assert res.shape == ref.shape and np.array_equal(res == 0, ref == 0)
    close(res, ref)
    if pname != "bkg_chid2_zscale_minmax":     # (a constant shift of one channel is normalised away again by zscale + minmax)
        assert not np.array_equal(ref[:, :, 0], ref[:, :, 1]) or not np.array_equal(ref[:, :, 1], ref[:, :, 2])   # channels really differ
