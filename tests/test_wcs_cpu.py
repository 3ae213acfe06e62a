"""caesar_yolo_amd.wcs.WCS against astropy.wcs.WCS(header) -- the object the reference builds at inference.py:473 and utils.py:236 / :411.
tests/golden/wcs.json holds astropy 4.3.1's world coordinates (oracle/gen_golden.py: gen_wcs) for the header kinds the reference's inputs
carry.  Tolerances: 1e-10 deg on world coordinates (measured worst 1.2e-13), 1e-6 px on the way back (measured 3e-8: astropy's own inverse
of the spherical rotation is what differs)."""
import json
import os

import numpy as np
import pytest

from caesar_yolo_amd.wcs import WCS, WCSError

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wcs.json")
with open(GOLD) as fp:
    CASES = json.load(fp)


def _lon_diff(a, b):
    return np.abs((np.asarray(a) - np.asarray(b) + 180.0) % 360.0 - 180.0)


@pytest.mark.parametrize("name", sorted(CASES))
def test_pixel_to_world_matches_astropy(name):
    c = CASES[name]
    w = WCS(c["header"])
    x, y = np.array(c["x"]), np.array(c["y"])
    for origin in (0, 1):
        a, d = w.all_pix2world(x, y, origin)
        ra, rd = c["world_%d" % origin]
        if w.is_celestial:
            assert _lon_diff(a, ra).max() <= 1e-10, name
        else:
            assert np.abs(a - np.array(ra)).max() <= 1e-10, name
        assert np.abs(d - np.array(rd)).max() <= 1e-10, name


@pytest.mark.parametrize("name", sorted(CASES))
def test_world_to_pixel_matches_astropy_and_round_trips(name):
    c = CASES[name]
    w = WCS(c["header"])
    x, y = np.array(c["x"]), np.array(c["y"])
    for origin in (0, 1):
        ra, rd = (np.array(v) for v in c["world_%d" % origin])
        px, py = w.wcs_world2pix(ra, rd, origin)
        bx, by = c["pix_back_%d" % origin]
        assert np.abs(px - np.array(bx)).max() <= 1e-6 and np.abs(py - np.array(by)).max() <= 1e-6, name
        assert np.abs(px - x).max() <= 1e-6 and np.abs(py - y).max() <= 1e-6, name


@pytest.mark.parametrize("name", sorted(CASES))
def test_pixel_scale_matrix(name):
    c = CASES[name]
    assert np.allclose(WCS(c["header"]).pixel_scale_matrix(), np.array(c["pixel_scale_matrix"]), rtol=0, atol=1e-15)


def test_scalar_input_and_fits_origin():
    w = WCS(CASES["tan"]["header"])
    a, d = w.wcs_pix2world(500.5, 400.5, 1)                    # the reference pixel maps to CRVAL
    assert abs(float(a) - 266.4) < 1e-12 and abs(float(d) + 28.9) < 1e-12
    a0, d0 = w.wcs_pix2world(499.5, 399.5, 0)
    assert float(a0) == float(a) and float(d0) == float(d)


def test_unsupported_projection_raises():
    h = dict(CASES["tan"]["header"])
    h["CTYPE1"], h["CTYPE2"] = "RA---ZPN", "DEC--ZPN"
    with pytest.raises(WCSError):
        WCS(h)


def test_sfinder_carries_the_wcs_of_its_header(tmp_path):
    """reference inference.py:473: self.wcs = WCS(header) once the image header has been read."""
    from caesar_yolo_amd import utils
    from caesar_yolo_amd.inference import wcs_of_header
    img = np.zeros((40, 50), np.float32)
    cards = dict(CASES["sin"]["header"])
    p = str(tmp_path / "w.fits")
    utils.write_fits_image(p, img, list(cards.items()))
    _, header = utils.read_fits_image(p)
    w = wcs_of_header(header)
    a, d = w.all_pix2world(np.array(CASES["sin"]["x"]), np.array(CASES["sin"]["y"]), 0)
    assert _lon_diff(a, CASES["sin"]["world_0"][0]).max() <= 1e-10
    assert np.abs(d - np.array(CASES["sin"]["world_0"][1])).max() <= 1e-10
    assert wcs_of_header({"CTYPE1": "RA---ZPN", "CTYPE2": "DEC--ZPN"}) is None          # logged and dropped, as utils.py:234-238
