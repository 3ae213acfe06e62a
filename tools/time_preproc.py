"""Time the preprocessing stage alone (cy_preproc: statistics + rejection checks + apply/pack) on a batch of tiles
(developer tool).  python tools/time_preproc.py [zscale+minmax|chan3+minmax|full] [tile] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import synth, utils, pipelines
from caesar_yolo_amd.model import YOLO
name = sys.argv[1] if len(sys.argv) > 1 else "chan3+minmax"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 640
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
specs = dict(pipelines.SPECS)
specs["full"] = [("bkg", dict(sigma=3)), ("shift", dict(sigma=1)), ("clip", dict(sigma_low=10, sigma_up=10)),
                 ("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))]
n = 8192
m = YOLO("seeded:n:5", precision="fp16", max_batch=B, max_imgsz=tile, device=0)
det = m.engine(0)
mos = det.mosaic_to_device(synth.make_mosaic(n, seed=20260105))
grid = [t for t in utils.generate_tiles(0, n - 1, 0, n - 1, tile, tile, 0.8, 0.8) if t[1] - t[0] == tile and t[3] - t[2] == tile][:B]
xy = [(t[0], t[2]) for t in grid]
cfg = pipelines.device_pipeline(specs[name]).program()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    reps = 5
    for _ in range(reps):
        det.preproc(mos, xy, tile, tile, tile, cfg)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    print("%s: %d tiles of %dx%d: %.3f ms per batch (%.1f us per tile)" % (name, len(xy), tile, tile, dt * 1e3, dt * 1e6 / len(xy)))
