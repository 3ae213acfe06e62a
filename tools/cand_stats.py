"""Candidate / detection count distribution over the S16k (or smaller) grid (developer tool)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import synth, utils, preprocessing as PP
from caesar_yolo_amd.model import YOLO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = YOLO("seeded:l:5", precision="fp16", max_batch=64, max_imgsz=512, device=0)
det = m.engine(0)
mos = det.mosaic_to_device(synth.make_mosaic(n, seed=20260104))
grid = [t for t in utils.generate_tiles(0, n - 1, 0, n - 1, 512, 512, 0.8, 0.8) if t[1] - t[0] == 512 and t[3] - t[2] == 512]
cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
cands, kept, merged = [], [], []
for i in range(0, len(grid), 64):
    ch = grid[i:i + 64]
    xy = [(t[0], t[2]) for t in ch]
    netin, st, lb = det.preproc(mos, xy, 512, 512, 512, cfg)
    pred = det.forward(netin)
    d, a, cnt = det.decode_nms(pred, 512, 512, 512, 512, 0.7, 0.5)
    cc = (C.c_int * len(ch))()
    det._chk(det.lib.cy_debug_cand_counts(det.ctx, cc, len(ch)))
    o, oc, _ = det.iou_merge(d, cnt, 0.7, 0.3, 0.8)
    cands += list(cc); kept += cnt.cpu().tolist(); merged += oc.cpu().tolist()
c = np.array(cands); k = np.array(kept); g = np.array(merged)
for name, v in (("candidates", c), ("nms kept", k), ("merged", g)):
    print(name, "mean %.1f median %d p90 %d p99 %d max %d" % (v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
