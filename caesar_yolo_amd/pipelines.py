"""Named preprocessing pipelines of the BASELINE configs as (stage, kwargs) lists in the order scripts/run.py builds them
(reference: scripts/run.py:272-302).  The same list feeds the device stage objects here and, in tests and bench.py's CPU
baseline, the oracle's `build_pipeline`."""
from . import preprocessing as PP

SPECS = {
    # test/run_inference.sh:6-10: --preprocessing --zscale_stretch --zscale_contrasts=0.25,0.25,0.25 --normalize_minmax 0..255
    "zscale+minmax": [("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))],
    # BASELINE config 5: --preprocessing --nchannels=3 --chan3_preproc --sigma_clip_baseline=0
    # --sigma_clip_low=10 --sigma_clip_up=10 --zscale_contrasts=0.25,0.25,0.25 --normalize_minmax --norm_min=0 --norm_max=255
    "chan3+minmax": [("chanresize", dict(nchans=3)),
                     ("chan3", dict(sigma_clip_baseline=0, sigma_clip_low=10, sigma_clip_up=10, zscale_contrast=0.25)),
                     ("minmax", dict(norm_min=0, norm_max=255))],
}

_STAGES = {"bkg": PP.BkgSubtractor, "shift": PP.SigmaClipShifter, "clip": PP.SigmaClipper, "chanresize": PP.ChanResizer,
           "zscale": PP.ZScaleTransformer, "chan3": PP.Chan3Trasformer, "minmax": PP.MinMaxNormalizer}


def device_pipeline(spec):
    """spec: a name in SPECS or a (stage, kwargs) list -> caesar_yolo_amd.preprocessing.DataPreprocessor."""
    if isinstance(spec, str):
        spec = SPECS[spec]
    return PP.DataPreprocessor([_STAGES[n](**kw) for n, kw in spec])
