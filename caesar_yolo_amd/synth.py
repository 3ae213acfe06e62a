"""Seeded synthetic radio mosaics for the benchmark and large parity tests ("S16k"/"S32k").

No survey data ships with the reference beyond one 132x132 cut-out (test/galaxy0001.fits) and the mosaic its parallel test
script names is absent, so the tiled configs run on these: Gaussian noise, elliptical-Gaussian compact sources, a few
extended ones, a NaN strip and an all-zero block (exercising the non-finite -> 0 ingest and the tile-rejection paths).
"""
import numpy as np


def make_mosaic(n=16384, seed=20260104, nsrc=None, next_=None, nan_strip=64, zero_block=512):
    rng = np.random.default_rng(seed)
    scale = (n / 16384.0) ** 2
    nsrc = int(20000 * scale) if nsrc is None else nsrc
    next_ = int(200 * scale) if next_ is None else next_
    img = rng.standard_normal((n, n), dtype=np.float32)
    img *= np.float32(1e-4)

    def stamp(cx, cy, smaj, q, pa, peak):
        r = int(np.ceil(5 * smaj))
        x0, x1 = max(0, int(cx) - r), min(n, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(n, int(cy) + r + 1)
        if x1 <= x0 or y1 <= y0:
            return
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        dx, dy = xx - np.float32(cx), yy - np.float32(cy)
        c, s = np.float32(np.cos(pa)), np.float32(np.sin(pa))
        u, v = dx * c + dy * s, -dx * s + dy * c
        img[y0:y1, x0:x1] += np.float32(peak) * np.exp(np.float32(-0.5) * ((u / np.float32(smaj)) ** 2 +
                                                                     (v / np.float32(smaj * q)) ** 2))
    cx, cy = rng.uniform(0, n, nsrc), rng.uniform(0, n, nsrc)
    sm, q, pa = rng.uniform(1.5, 6.0, nsrc), rng.uniform(0.5, 1.0, nsrc), rng.uniform(0, np.pi, nsrc)
    pk = 10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2), nsrc)
    for i in range(nsrc):
        stamp(cx[i], cy[i], sm[i], q[i], pa[i], pk[i])
    cx, cy = rng.uniform(0, n, next_), rng.uniform(0, n, next_)
    sm, q, pa = rng.uniform(10, 40, next_), rng.uniform(0.5, 1.0, next_), rng.uniform(0, np.pi, next_)
    pk = 10 ** rng.uniform(np.log10(5e-4), np.log10(5e-3), next_)
    for i in range(next_):
        stamp(cx[i], cy[i], sm[i], q[i], pa[i], pk[i])
    if nan_strip:
        img[:, n - nan_strip:] = np.nan
    if zero_block and n >= 4 * zero_block:
        img[2 * zero_block:3 * zero_block, 2 * zero_block:3 * zero_block] = 0.0
    return img


FITS_CARDS = [("BUNIT", "Jy/beam"), ("BMAJ", 0.0026), ("BMIN", 0.0021), ("BPA", 84.0), ("CDELT1", -0.0005), ("CDELT2", 0.0005)]
