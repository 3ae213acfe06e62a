"""Where the ingest time of the benchmark goes (developer tool): FITS header + map, partition, region uploads.
python tools/time_ingest.py [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import synth, utils, pipelines as CP
from caesar_yolo_amd.model import YOLO
from caesar_yolo_amd.inference import TileEngine, MosaicSource
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
path = "/tmp/cy_ingest_%d.fits" % size
if not os.path.exists(path):
    utils.write_fits_image(path, synth.make_mosaic(size, seed=20260104), synth.FITS_CARDS)
m = YOLO("seeded:n:5", precision="fp16", max_batch=256, max_imgsz=512, device=0)
det = m.engine(0)
grid = utils.generate_tiles(0, size - 1, 0, size - 1, 512, 512, 0.8, 0.8)
cfg = CP.device_pipeline("zscale+minmax").program()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    data, _ = utils.read_fits_image(path)
    t1 = time.time()
    src = MosaicSource(data, big_endian=True)
    orig = src.region
    def timed(det_, *a):
        torch.cuda.synchronize(); s = time.time(); r = orig(det_, *a); torch.cuda.synchronize()
        print("   region x %d..%d y %d..%d: %.1f ms" % (a + (1e3 * (time.time() - s),)))
        return r
    src.region = timed
    eng = TileEngine(det, src, grid, cfg, 512, 0.7, 0.5, 0.3, 0.8, 0, 1, 256)
    torch.cuda.synchronize(); t2 = time.time()
    print("pass %d: read_fits_image %.1f ms, TileEngine (partition + uploads) %.1f ms, total %.1f ms, %.0f MB" % (
        it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0), src.bytes_uploaded / 1e6))
    del eng, src
