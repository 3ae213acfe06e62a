"""The N>1 path on CPU: two ranks over gloo shard a tile grid, each runs its share through a stand-in detector that
replays recorded per-tile detections (the per-tile IoU merge done by the oracle), ONE all-gather moves the
fixed-capacity records, rank 0 merges.  The catalog must equal, byte for byte, the one the reference wrote for the same
detections (tests/golden/catalog_tiled_*.json) -- i.e. it does not depend on the number of ranks."""
import json
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "spurious", 1: "compact", 2: "extended", 3: "extended-multisland", 4: "flagged"}


class ReplayDetector(object):
    """Same `detect_tiles` contract as caesar_yolo_amd.model.HipDetector, on CPU tensors."""
    max_batch = 8
    tdev = torch.device("cpu")

    def __init__(self, fx):
        from oracle import postproc_ref as R
        self.R, self.fx = R, fx
        self.by_origin = {(t[0], t[2]): i for i, t in enumerate(fx["grid"])}
        self.skipped = {c[0] for c in fx["calls"] if c[1] < 0}

    def detect_tiles(self, mosaic, xy, th, tw, imgsz, cfg, conf, iou, soft, hard, out=None, flush=True):
        det, cnt, status = out
        det.zero_()
        for b, o in enumerate(xy):
            tid = self.by_origin[tuple(o)]
            if tid in self.skipped:
                status[b], cnt[b] = 2, 0
                continue
            bx, s, c = (np.array(x, np.float32) for x in self.fx["dets"][tid])
            kb, ks, kc, _ = self.R.process_detections(bx, s, c, conf, soft, hard)
            n = len(ks)
            status[b], cnt[b] = 0, n
            if n:
                det[b, :n, :4] = torch.from_numpy(kb)
                det[b, :n, 4] = torch.from_numpy(ks)
                det[b, :n, 5] = torch.from_numpy(kc.astype(np.float32))
        return det, cnt, status


def _worker(rank, world, tag, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from caesar_yolo_amd.inference import TileEngine
    fx = json.load(open(os.path.join(ROOT, "tests/golden/catalog_tiled_%s.json" % tag)))
    c = fx["config"]
    eng = TileEngine(ReplayDetector(fx), None, fx["grid"], None, c["img_size"], c["score_thr"], c["iou_thr"], c["soft"],
                     c["hard"], rank, world, batch=5)
    n = eng.run_local()
    eng.gather()
    if rank == 0:
        src, stats = eng.catalog(NAMES)
        q.put((json.dumps({"sources": src}, indent=2, sort_keys=True), stats, n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tag,world", [("a", 2), ("c", 2), ("c", 3), ("d", 2), ("d", 3)])
def test_two_rank_catalog_equals_reference(tag, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, tag, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    text, stats, n0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = json.load(open(os.path.join(ROOT, "tests/golden/catalog_tiled_%s.json" % tag)))
    assert text == fx["catalog_text"]
    assert stats["tiles"] == len(fx["grid"]) and stats["skipped"] == sum(1 for c in fx["calls"] if c[1] < 0)
    assert 0 < n0 < len(fx["grid"])          # rank 0 really processed only its share


def test_single_rank_catalog_equals_reference():
    from caesar_yolo_amd.inference import TileEngine
    fx = json.load(open(os.path.join(ROOT, "tests/golden/catalog_tiled_b.json")))
    c = fx["config"]
    eng = TileEngine(ReplayDetector(fx), None, fx["grid"], None, c["img_size"], c["score_thr"], c["iou_thr"], c["soft"],
                     c["hard"], 0, 1, batch=7)
    eng.run_local()
    eng.gather()
    src, _ = eng.catalog(NAMES)
    assert json.dumps({"sources": src}, indent=2, sort_keys=True) == fx["catalog_text"]


def test_shard_covers_everything_once():
    from caesar_yolo_amd.inference import shard
    items = list(range(1601))
    for w in (1, 2, 3, 4, 8):
        parts = [shard(items, r, w) for r in range(w)]
        assert sum(parts, []) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
