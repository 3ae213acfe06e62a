"""Time every rank's LOCAL share of the S16k grid at a given world size, one after the other on one GPU (developer tool):
the N-GPU step time is max over ranks of this + gather + cross-tile merge.  python tools/emulate_ranks.py [world] [batch] [size] [rank,rank,...]
EMU_PG=1: a one-rank RCCL process group is initialised (eagerly, BEFORE the detector context exists) and every pass ends with the
record all-gather through it -- the collective library's stream then lives beside the context's four, as in a real N > 1 run
(GPU_MAX_HW_QUEUES=4 restores the HIP runtime's default queue count: the case that serialised two streams)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import synth, utils, preprocessing as PP
from caesar_yolo_amd.model import YOLO
from caesar_yolo_amd.inference import TileEngine
if os.environ.get("EMU_PG") == "1":
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 192
size = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
m = YOLO("seeded:l:5", precision="fp16", max_batch=batch, max_imgsz=512, device=0)
det = m.engine(0)
mos = det.mosaic_to_device(synth.make_mosaic(size, seed=20260104))
grid = utils.generate_tiles(0, size - 1, 0, size - 1, 512, 512, 0.8, 0.8)
cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
only = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(world))      # ranks to time (default: all)
worst = 0.0
for r in only:
    eng = TileEngine(det, mos, grid, cfg, 512, 0.7, 0.5, 0.3, 0.8, r, world, batch)
    for _ in range(2):
        eng.run_local()
    torch.cuda.synchronize()
    reps = 5
    t = time.time()
    for _ in range(reps):
        eng.run_local()
        torch.cuda.synchronize()
        if os.environ.get("EMU_PG") == "1":          # this rank's record buffer through RCCL (a group of one: the engine's world is emulated)
            out1 = torch.empty((1,) + tuple(eng.rec.shape), dtype=eng.rec.dtype, device=eng.rec.device)
            dist.all_gather_into_tensor(out1, eng.rec)
            torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    worst = max(worst, dt)
    print("rank %d: %4d tiles, batches %s: %.2f ms (%.0f tiles/s)" % (r, eng.n_my, [(p[0], p[1], p[2]) for p in eng.plan], dt * 1e3, eng.n_my / dt))
print("max over ranks %.2f ms -> %.0f tiles/s for %d tiles before gather+merge" % (worst * 1e3, len(grid) / worst, len(grid)))
