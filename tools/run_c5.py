"""BASELINE config 5 shape at reduced size: --chan3_preproc 3-channel, 640x640 tiles step 0.8 (developer check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import synth, utils, preprocessing as PP
from caesar_yolo_amd.model import YOLO
from caesar_yolo_amd.inference import TileEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = YOLO("seeded:l:5", precision="fp16", max_batch=96, max_imgsz=640, device=0)
det = m.engine(0)
mos = det.mosaic_to_device(synth.make_mosaic(n, seed=20260105))
grid = utils.generate_tiles(0, n - 1, 0, n - 1, 640, 640, 0.8, 0.8)
cfg = PP.DataPreprocessor([PP.ChanResizer(3), PP.Chan3Trasformer(0, 10, 10, 0.25), PP.MinMaxNormalizer(0, 255)]).program()
eng = TileEngine(det, mos, grid, cfg, 640, 0.7, 0.5, 0.3, 0.8, 0, 1, 96)
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    eng.run_local(); eng.gather(); rec, st = eng.merged_records()
    torch.cuda.synchronize(); dt = time.time() - t
    print("pass %d: %d tiles (%s) in %.3f s = %.1f tiles/s; %d sources, stats %s" % (it, len(grid), sorted(set((t[3]-t[2], t[1]-t[0]) for t in grid)), dt, len(grid) / dt, len(rec), st))
