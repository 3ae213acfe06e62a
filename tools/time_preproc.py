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
names = (sys.argv[1] if len(sys.argv) > 1 else "chan3+minmax").split(",")
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 640
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
specs = dict(pipelines.SPECS)
specs["full"] = [("bkg", dict(sigma=3)), ("shift", dict(sigma=1)), ("clip", dict(sigma_low=10, sigma_up=10)),
                 ("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))]
specs["clip10"] = [("clip", dict(sigma_low=10, sigma_up=10))]
specs["clip3"] = [("clip", dict(sigma_low=0, sigma_up=10))]
specs["bkg"] = [("bkg", dict(sigma=3))]
specs["zscale"] = [("zscale", dict(contrasts=[0.25] * 3))]
specs["minmax"] = [("minmax", dict(norm_min=0, norm_max=255))]
specs["none"] = []
specs["clip3z"] = [("clip", dict(sigma_low=0, sigma_up=10)), ("zscale", dict(contrasts=[0.25] * 3))]
specs["clip3zm"] = specs["clip3z"] + [("minmax", dict(norm_min=0, norm_max=255))]
specs["chan3"] = specs["chan3+minmax"][:2]
n = 8192
m = YOLO("seeded:n:5", precision="fp16", max_batch=B, max_imgsz=tile, device=0)
det = m.engine(0)
mos = det.mosaic_to_device(synth.make_mosaic(n, seed=20260105))
grid = [t for t in utils.generate_tiles(0, n - 1, 0, n - 1, tile, tile, 0.8, 0.8) if t[1] - t[0] == tile and t[3] - t[2] == tile][:B]
xy = [(t[0], t[2]) for t in grid]
for name in names:
    cfg = pipelines.device_pipeline(specs[name]).program()
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        reps = 5
        for _ in range(reps):
            det.preproc(mos, xy, tile, tile, tile, cfg)
        torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    if os.environ.get("CY_STAMPS"):
        import ctypes
        st = (ctypes.c_ulonglong * 8)()
        det.lib.cy_debug_stamps(st, 3)
        tot = float(sum(st)) or 1.0
        print("   stamps (share of stamped cycles): moments %.2f, radix median %.2f, bracket select %.2f, clip-run total %.2f, zscale %.2f, histeq %.2f, minmax %.2f"
              % (st[0] / tot, st[1] / tot, st[2] / tot, st[3] / tot, st[4] / tot, st[5] / tot, st[6] / tot), [int(v) for v in st])
    c = det.counters(reset=True)
    print("%s: %d tiles of %dx%d: %.3f ms per batch (%.1f us per tile); median bracket hits %d, misses %d" % (
        name, len(xy), tile, tile, dt * 1e3, dt * 1e6 / len(xy), c["median_bracket_hits"], c["median_bracket_misses"]))
