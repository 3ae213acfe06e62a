"""CPU oracle for the per-tile preprocessing path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (caesar_yolo_amd) never does.

Restates, in plain numpy, the reference stages reachable from scripts/run.py and the
third-party routines they call.  Every function cites what it follows:

  reference (relative to /root/reference):
    caesar_yolo/preprocessing.py  MinMaxNormalizer :75-111, BkgSubtractor :591-658,
        SigmaClipShifter :664-717, SigmaClipper :723-771, ZScaleTransformer :934-971,
        HistEqualizer :977-1012, Chan3Trasformer :1020-1072, ChanResizer :1077-1133,
        DataPreprocessor :47-67
    caesar_yolo/evaluation.py :146-154 (2-D -> 3-channel float64 cube), :171-176 (row check)
  third party (absent from /root/reference; versions of the build container's conda env):
    astropy 4.3.1  visualization/interval.py:229-296 (ZScaleInterval.get_limits), :66-80 (__call__)
                   stats/sigma_clipping.py:385-433 (_sigmaclip_noaxis), :288-296 (_compute_bounds),
                   :815-937 (sigma_clipped_stats)
    scikit-image 0.18.3  exposure/exposure.py:78-134 (histogram), :181-223 (equalize_hist)

Pinned by tests/golden/preproc.npz, produced by oracle/gen_golden.py from the imported
reference (test: tests/test_oracle_preproc.py).  Statistics that the reference computes
with bottleneck's sequential sums are computed here with numpy's pairwise sums: agreement
is to ~1e-15 relative, not bit-for-bit, and the tests say so.
"""
import numpy as np

ZS_NSAMPLES = 1000
ZS_MAX_REJECT = 0.5
ZS_MIN_NPIX = 5
ZS_KREJ = 2.5
ZS_MAX_ITER = 5


def nonzero_finite(x):
    """cond = (x != 0) & isfinite(x): the mask every stage uses (preprocessing.py:99, 606, 677, 737, 950)."""
    return np.logical_and(x != 0, np.isfinite(x))


# --------------------------------------------------------------------------- astropy restatements
def zscale_limits(values, contrast=0.25):
    """astropy ZScaleInterval.get_limits (interval.py:229-296)."""
    values = np.asarray(values)
    values = values[np.isfinite(values)]          # row-major flatten; zeros ARE included
    stride = int(max(1.0, values.size / ZS_NSAMPLES))
    samples = values[::stride][:ZS_NSAMPLES].copy()
    samples.sort()
    npix = len(samples)
    vmin, vmax = samples[0], samples[-1]
    minpix = max(ZS_MIN_NPIX, int(npix * ZS_MAX_REJECT))
    x = np.arange(npix)
    ngoodpix, last_ngoodpix = npix, npix + 1
    badpix = np.zeros(npix, dtype=bool)
    ngrow = max(1, int(npix * 0.01))
    kernel = np.ones(ngrow, dtype=bool)
    fit = None
    for _ in range(ZS_MAX_ITER):
        if ngoodpix >= last_ngoodpix or ngoodpix < minpix:
            break
        fit = np.polyfit(x, samples, deg=1, w=(~badpix).astype(int))
        flat = samples - np.poly1d(fit)(x)
        threshold = ZS_KREJ * flat[~badpix].std()
        badpix[(flat < -threshold) | (flat > threshold)] = True
        badpix = np.convolve(badpix, kernel, mode="same")
        last_ngoodpix = ngoodpix
        ngoodpix = np.sum(~badpix)
    if ngoodpix >= minpix:
        slope = fit[0]
        if contrast > 0:
            slope = slope / contrast
        center_pixel = (npix - 1) // 2
        median = np.median(samples)
        vmin = max(vmin, median - (center_pixel - 1) * slope)
        vmax = min(vmax, median + (npix - center_pixel) * slope)
    return float(vmin), float(vmax)


def zscale_apply(values, vmin, vmax):
    """astropy BaseInterval.__call__ with clip=True (interval.py:66-80)."""
    out = np.subtract(values, float(vmin))
    if (vmax - vmin) != 0:
        np.true_divide(out, vmax - vmin, out=out)
    np.clip(out, 0.0, 1.0, out=out)
    return out


def sigma_clip_1d(data_1d, sigma_lower, sigma_upper, sigma=3.0, maxiters=5):
    """astropy SigmaClip._sigmaclip_noaxis (sigma_clipping.py:385-433).

    `sigma_lower or sigma`: a 0 falls back to `sigma` (sigma_clipping.py:226-227).
    Returns (surviving values, last lower bound, last upper bound)."""
    lo_s = sigma_lower or sigma
    up_s = sigma_upper or sigma
    filtered = np.asarray(data_1d, dtype=np.float64).ravel()
    filtered = filtered[np.isfinite(filtered)]
    nchanged, it = 1, 0
    lo = hi = np.nan
    while nchanged != 0 and it < maxiters:
        it += 1
        size = filtered.size
        cen = np.median(filtered)
        std = np.std(filtered)
        lo = cen - std * lo_s
        hi = cen + std * up_s
        filtered = filtered[(filtered >= lo) & (filtered <= hi)]
        nchanged = size - filtered.size
    return filtered, float(lo), float(hi)


def sigma_clipped_stats_1d(data_1d, sigma=3.0):
    """astropy sigma_clipped_stats (sigma_clipping.py:815-937): mean, median, std(ddof 0) of survivors."""
    f, _, _ = sigma_clip_1d(data_1d, None, None, sigma=sigma)
    return float(np.mean(f)), float(np.median(f)), float(np.std(f))


def equalize_hist(image, nbins=256):
    """skimage.exposure.equalize_hist for a float image (exposure.py:78-134, 181-223)."""
    flat = image.flatten()
    hist, edges = np.histogram(flat, bins=nbins, range=None)
    centers = (edges[:-1] + edges[1:]) / 2.0
    cdf = hist.cumsum()
    cdf = cdf / float(cdf[-1])
    return np.interp(image.flat, centers, cdf).reshape(image.shape)


# --------------------------------------------------------------------------- reference stages
class MinMaxNormalizer(object):
    """preprocessing.py:75-111."""

    def __init__(self, norm_min=0, norm_max=1):
        self.norm_min, self.norm_max = norm_min, norm_max

    def __call__(self, data):
        if data is None:
            return None
        out = np.copy(data)
        for i in range(data.shape[-1]):
            ch = data[:, :, i]
            cond = nonzero_finite(ch)
            v = ch[cond]
            if v.size == 0:
                return None
            mn, mx = v.min(), v.max()
            n = (ch - mn) / (mx - mn) * (self.norm_max - self.norm_min) + self.norm_min
            n[~cond] = 0
            out[:, :, i] = n
        return out


class BkgSubtractor(object):
    """preprocessing.py:591-658."""

    def __init__(self, sigma=3, use_mask_box=False, mask_fract=0.7, chid=-1):
        self.sigma, self.use_mask_box, self.mask_fract, self.chid = sigma, use_mask_box, mask_fract, chid

    def _sub(self, data):
        cond = nonzero_finite(data)
        bkgdata = np.copy(data)
        if self.use_mask_box:
            sh = data.shape
            xc, yc = int(sh[1] / 2), int(sh[0] / 2)
            dy, dx = int(sh[0] * self.mask_fract / 2.0), int(sh[1] * self.mask_fract / 2.0)
            bkgdata[yc - dy:yc + dy, xc - dx:xc + dx] = 0
        v = bkgdata[nonzero_finite(bkgdata)]
        bkgval, _, _ = sigma_clipped_stats_1d(v, sigma=self.sigma)
        out = data - bkgval
        out[~cond] = 0
        return out

    def __call__(self, data):
        if data is None:
            return None
        out = np.copy(data)
        for i in range(data.shape[-1]):
            if self.chid != -1 and i != self.chid:
                continue
            out[:, :, i] = self._sub(data[:, :, i])
        return out


class SigmaClipShifter(object):
    """preprocessing.py:664-717."""

    def __init__(self, sigma=1.0, chid=-1):
        self.sigma, self.chid = sigma, chid

    def _clip(self, data):
        cond = nonzero_finite(data)
        clipmean, _, std = sigma_clipped_stats_1d(data[cond], sigma=self.sigma)
        newzero = clipmean + self.sigma * std
        out = np.copy(data)
        out -= newzero
        out[out < 0] = 0
        out[~cond] = 0
        return out

    def __call__(self, data):
        if data is None:
            return None
        out = np.copy(data)
        for i in range(data.shape[-1]):
            if self.chid != -1 and i != self.chid:
                continue
            out[:, :, i] = self._clip(data[:, :, i])
        return out


class SigmaClipper(object):
    """preprocessing.py:723-771."""

    def __init__(self, sigma_low=10.0, sigma_up=10.0, chid=-1):
        self.sigma_low, self.sigma_up, self.chid = sigma_low, sigma_up, chid

    def _clip(self, data):
        cond = nonzero_finite(data)
        _, lo, hi = sigma_clip_1d(data[cond], self.sigma_low, self.sigma_up)
        out = np.copy(data)
        out[out < lo] = lo
        out[out > hi] = hi
        out[~cond] = 0
        return out

    def __call__(self, data):
        if data is None:
            return None
        out = np.copy(data)
        for i in range(data.shape[-1]):
            if self.chid != -1 and i != self.chid:
                continue
            out[:, :, i] = self._clip(data[:, :, i])
        return out


class ChanResizer(object):
    """preprocessing.py:1077-1133 (replicate last channel when expanding)."""

    def __init__(self, nchans):
        self.nchans = nchans

    def __call__(self, data):
        if data is None or self.nchans > 1000 or self.nchans <= 0:
            return None
        cur = 1 if data.ndim == 2 else data.shape[-1]
        if self.nchans == cur:
            return data
        if data.ndim == 2:
            data = np.expand_dims(data, axis=-1)
        out = np.zeros((data.shape[0], data.shape[1], self.nchans))
        for i in range(self.nchans):
            out[:, :, i] = data[:, :, i] if i < cur else data[:, :, cur - 1]
        return out


class ZScaleTransformer(object):
    """preprocessing.py:934-971."""

    def __init__(self, contrasts=(0.25, 0.25, 0.25)):
        self.contrasts = list(contrasts)

    def __call__(self, data):
        if data is None:
            return None
        cond = nonzero_finite(data)
        if len(self.contrasts) < data.shape[-1]:
            return None
        out = np.copy(data)
        for i in range(data.shape[-1]):
            ch = out[:, :, i]
            vmin, vmax = zscale_limits(ch, self.contrasts[i])
            out[:, :, i] = zscale_apply(ch, vmin, vmax)
        out[~cond] = 0
        return out


class HistEqualizer(object):
    """preprocessing.py:977-1012 (adaptive=False branch, the only one Chan3Trasformer uses)."""

    def __call__(self, data):
        if data is None:
            return None
        cond = nonzero_finite(data)
        out = np.copy(data)
        for i in range(data.shape[-1]):
            out[:, :, i] = equalize_hist(data[:, :, i])
        out[~cond] = 0
        return out


class Chan3Trasformer(object):
    """preprocessing.py:1020-1072 (name spelled as in the reference)."""

    def __init__(self, sigma_clip_baseline=0, sigma_clip_low=1, sigma_clip_up=20, zscale_contrast=0.25):
        self.b, self.lo, self.up, self.c = sigma_clip_baseline, sigma_clip_low, sigma_clip_up, zscale_contrast

    def __call__(self, data):
        if data is None:
            return None
        cubed = ChanResizer(3)(data)
        cubed = np.copy(cubed)
        sc1 = SigmaClipper(self.b, self.up)
        sc2 = SigmaClipper(self.lo, self.up)
        zs = ZScaleTransformer([self.c])
        he = HistEqualizer()
        cubed[:, :, 0] = zs(sc1(np.expand_dims(cubed[:, :, 0], -1)))[:, :, 0]
        cubed[:, :, 1] = zs(sc2(np.expand_dims(cubed[:, :, 1], -1)))[:, :, 0]
        cubed[:, :, 2] = he(np.expand_dims(cubed[:, :, 2], -1))[:, :, 0]
        return cubed


class DataPreprocessor(object):
    """preprocessing.py:47-67: stages applied first-to-last."""

    def __init__(self, stages):
        self.stages = list(stages)

    def __call__(self, data):
        for s in self.stages:
            data = s(data)
        return data


def to_cube(img2d):
    """evaluation.py:146-154: 2-D image -> (H,W,3) float64, three identical channels."""
    c = np.zeros((img2d.shape[0], img2d.shape[1], 3))
    for i in range(3):
        c[:, :, i] = img2d
    return c


def rows_constant(image_hwc):
    """evaluation.py:171-176: loops i over the 3 channels but indexes image[i] = ROW i.
    True => the reference returns -1 and the tile is skipped."""
    for i in range(image_hwc.shape[-1]):
        if np.min(image_hwc[i]) == np.max(image_hwc[i]):
            return True
    return False


def build_pipeline(spec):
    """spec: list of (stage_name, kwargs) in scripts/run.py:274-293 order."""
    table = {"bkg": BkgSubtractor, "shift": SigmaClipShifter, "clip": SigmaClipper, "chanresize": ChanResizer,
             "zscale": ZScaleTransformer, "chan3": Chan3Trasformer, "minmax": MinMaxNormalizer}
    return DataPreprocessor([table[n](**kw) for n, kw in spec])
