"""Host-side mirror of the reference's preprocessing interface (caesar_yolo/preprocessing.py).

Same class names, constructor arguments and composition rule (`DataPreprocessor(stages)`, first stage applied first,
scripts/run.py:272-302), but the objects carry no numpy arithmetic: they describe the stage program that the HIP
kernels (csrc/cy_preproc.hip) execute per tile in HBM.  `DataPreprocessor.program()` lowers the stage list to the
C-ABI `cy_preproc_cfg` (one op list per output channel).

Only the stages reachable from the reference CLI exist here: BkgSubtractor, SigmaClipShifter,
SigmaClipper, ChanResizer, ZScaleTransformer, Chan3Trasformer (sic), MinMaxNormalizer.  Per-channel stage selection
(`chid`, from --bkg_chid / --clip_chid, scripts/run.py:89, :98, :275-281) lowers the stage into that channel's program
only: the reference skips the other channels (caesar_yolo/preprocessing.py:653, :712, :766).
"""
from . import lib as L


class _Stage(object):
    def lower(self, progs):
        raise NotImplementedError


def _targets(progs, chid):
    """The channel programs a stage with this `chid` is appended to (-1: all three)."""
    chid = int(chid)
    if chid == -1:
        return progs
    if chid not in (0, 1, 2):
        # the reference would simply never match `i == chid` and the stage would do nothing; say so instead
        raise ValueError("chid must be -1 or a channel index 0..2 (got %r)" % (chid,))
    return [progs[chid]]


class BkgSubtractor(_Stage):
    """caesar_yolo/preprocessing.py:591-658"""

    def __init__(self, sigma=3, use_mask_box=False, mask_fract=0.7, chid=-1, **kw):
        self.sigma, self.use_mask_box, self.mask_fract, self.chid = float(sigma), bool(use_mask_box), float(mask_fract), int(chid)
        _targets([[], [], []], chid)

    def lower(self, progs):
        for p in _targets(progs, self.chid):
            p.append((L.OP_BKG, self.sigma, self.mask_fract, 0.0, int(self.use_mask_box)))


class SigmaClipShifter(_Stage):
    """caesar_yolo/preprocessing.py:664-717"""

    def __init__(self, sigma=1.0, chid=-1, **kw):
        self.sigma, self.chid = float(sigma), int(chid)
        _targets([[], [], []], chid)

    def lower(self, progs):
        for p in _targets(progs, self.chid):
            p.append((L.OP_SHIFT, self.sigma, 0.0, 0.0, 0))


class SigmaClipper(_Stage):
    """caesar_yolo/preprocessing.py:723-771"""

    def __init__(self, sigma_low=10.0, sigma_up=10.0, chid=-1, **kw):
        self.sigma_low, self.sigma_up, self.chid = float(sigma_low), float(sigma_up), int(chid)
        _targets([[], [], []], chid)

    def lower(self, progs):
        for p in _targets(progs, self.chid):
            p.append((L.OP_CLIP, self.sigma_low, self.sigma_up, 0.0, 0))


class ChanResizer(_Stage):
    """caesar_yolo/preprocessing.py:1077-1133.  Analyzer.predict has already replicated the tile to 3 channels
    (caesar_yolo/evaluation.py:146-154), so nchans=3 is the identity; anything else cannot feed the detector."""

    def __init__(self, nchans, **kw):
        if nchans != 3:
            raise NotImplementedError("the detector takes 3 channels; ChanResizer(nchans=%r) is not supported" % nchans)
        self.nchans = nchans

    def lower(self, progs):
        pass


class ZScaleTransformer(_Stage):
    """caesar_yolo/preprocessing.py:934-971"""

    def __init__(self, contrasts=(0.25, 0.25, 0.25), **kw):
        self.contrasts = [float(c) for c in contrasts]

    def lower(self, progs):
        if len(self.contrasts) < len(progs):
            raise ValueError("Invalid constrasts given (constrast list size=%d < nchans=%d)" % (len(self.contrasts), len(progs)))
        for i, p in enumerate(progs):
            p.append((L.OP_ZSCALE, self.contrasts[i], 0.0, 0.0, 0))


class MinMaxNormalizer(_Stage):
    """caesar_yolo/preprocessing.py:75-111"""

    def __init__(self, norm_min=0, norm_max=1, **kw):
        self.norm_min, self.norm_max = float(norm_min), float(norm_max)

    def lower(self, progs):
        for p in progs:
            p.append((L.OP_MINMAX, self.norm_min, self.norm_max, 0.0, 0))


class Chan3Trasformer(_Stage):
    """caesar_yolo/preprocessing.py:1020-1072: ch1 = zscale(sigmaclip(baseline, up)), ch2 = zscale(sigmaclip(low, up)),
    ch3 = histogram equalisation."""

    def __init__(self, sigma_clip_baseline=0, sigma_clip_low=1, sigma_clip_up=20, zscale_contrast=0.25, **kw):
        self.b, self.lo, self.up, self.c = float(sigma_clip_baseline), float(sigma_clip_low), float(sigma_clip_up), float(zscale_contrast)

    def lower(self, progs):
        progs[0].append((L.OP_CLIP, self.b, self.up, 0.0, 0))
        progs[0].append((L.OP_ZSCALE, self.c, 0.0, 0.0, 0))
        progs[1].append((L.OP_CLIP, self.lo, self.up, 0.0, 0))
        progs[1].append((L.OP_ZSCALE, self.c, 0.0, 0.0, 0))
        progs[2].append((L.OP_HISTEQ, 0.0, 0.0, 0.0, 0))


class DataPreprocessor(object):
    """caesar_yolo/preprocessing.py:47-67.  Not callable on arrays: the pipeline runs on the GPU per tile."""

    def __init__(self, stages):
        self.stages = list(stages)

    def program(self):
        """-> lib.cy_preproc_cfg.  Identical channel programs collapse to one (the reference computes every statistic
        three times on identical channels)."""
        progs = [[], [], []]
        for s in self.stages:
            s.lower(progs)
        for p in progs:
            if len(p) > L.CY_MAX_STAGES:
                raise ValueError("more than %d preprocessing stages per channel" % L.CY_MAX_STAGES)
        same = progs[0] == progs[1] == progs[2]
        cfg = L.cy_preproc_cfg()
        cfg.nprog = 1 if same else 3
        for i in range(1 if same else 3):
            cfg.prog[i].n = len(progs[i])
            for j, (op, p0, p1, p2, flag) in enumerate(progs[i]):
                st = cfg.prog[i].st[j]
                st.op, st.p0, st.p1, st.p2, st.flag = op, p0, p1, p2, flag
        if same and len(progs[0]) == 0:
            cfg.nprog = 0
        return cfg

    def __call__(self, data):
        raise RuntimeError("DataPreprocessor runs inside the HIP tile pipeline (cy_preproc); it has no CPU implementation")


def no_preprocessing():
    cfg = L.cy_preproc_cfg()
    cfg.nprog = 0
    return cfg
