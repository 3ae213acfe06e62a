"""Tile scheduler + catalog: host mirror of the reference's `SFinder` (caesar_yolo/inference.py:280-1288) for the
MI355X path.

Same public surface -- `SFinder(model, config).run()` (single frame, :485-552) and `.run_parallel()` (tiled, :578-658),
int status returns, `catalog_<image_id>.json` / `out_<image_id>.json` outputs -- but a different engine:

  * the mosaic is read once, lives in HBM, and tiles are cropped on device (reference: every tile re-opens the file,
    caesar_yolo/inference.py:190-195);
  * tiles are grouped by shape and run through the whole per-tile path in batches (`cy_detect_tiles`); the reference
    runs batch 1 sequentially (:611-622);
  * ranks are one process per GPU under torch.distributed; tile t of a shape class goes to rank (contiguous chunks),
    and the only exchange is ONE all-gather of fixed-capacity detection records (replaces the pickled send/recv of
    :936-984).  Tile geometry is recomputed identically on every rank, so only detections travel;
  * the cross-tile merge (:663-726, :731-931) runs in the library's host code on tile-id-ordered records, which makes
    the catalog independent of the number of GPUs (the reference's source numbering depends on the MPI rank count,
    SURVEY.md section 8 a12; this build fixes the order to the 1-process order).
"""
import ctypes as C
import json
import logging
import os
import time
import numpy as np
import torch

from . import lib as L
from . import utils
from .preprocessing import no_preprocessing

logger = logging.getLogger("caesar_yolo_amd")


def wcs_of_header(header):
    """The `WCS(header)` of caesar_yolo/inference.py:473 (and utils.py:234-238, where a failure is logged and the WCS dropped):
    caesar_yolo_amd.wcs restates the celestial part of the standard; nothing on the detect path reads it."""
    from .wcs import WCS
    try:
        return WCS(header)
    except Exception as e:
        logger.warning("Failed to get wcs from header (err=%s)!" % str(e))
        return None


def shard(items, rank, world):
    """Contiguous chunk `rank` of `world` (sizes differ by at most one)."""
    n = len(items)
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return items[lo:hi]


# Cost model of one rank's pass, in units of "one full-size tile": a batch of n tiles of relative letterboxed area w costs
# n*w plus a fixed launch/pipeline overhead per batch (measured on MI355X at world 8: 198 full tiles in one batch 28.7 ms,
# in two 30.5 ms, in three 36.6 ms; full rate 7285 tiles/s -> 12-25 tile-equivalents per batch; small batches of
# ragged tiles run well below the full rate).
# Round 4 refit on the emulation of N = 2 / 4 / 8 (tools/emulate_ranks.py; 0.1008 ms per tile-equivalent): a rank's FIRST batch pays its
# preprocessing and the pipeline fill in full, later batches run behind the previous one's forward pass, and a small batch (ragged edge
# tiles, or a few full tiles) runs its deep layers far below the full rate wherever it runs (the small-batch lane shares the GPU with
# the main one).  Residuals of the fit: -3 .. +3 % of a rank's time over the seven measured ranks.
BATCH_OVERHEAD_FIRST = 23.0
BATCH_OVERHEAD_NEXT = 11.5
SMALL_BATCH_EXTRA_TILES = 20.0   # minus 1.0 per full-size batch of the same rank to run beside (up to three)
BATCH_OVERHEAD_TILES = BATCH_OVERHEAD_FIRST      # (name kept for tools that print the model)


def _rank_cost(segments, cost, batch):
    """segments: list of (shape, n tiles).  Estimated pass time in tile-equivalents."""
    t, nb = 0.0, 0
    nbig = sum((n + batch - 1) // batch for _, n in segments if n >= 128)
    for shp, n in segments:
        if n:
            k = (n + batch - 1) // batch
            t += n * cost[shp]
            nb += k
            if n < 64:                                 # the deep layers do not fill the chip: below the linear model
                t += SMALL_BATCH_EXTRA_TILES - 1.0 * min(nbig, 3)
    if nb:
        t += BATCH_OVERHEAD_FIRST + BATCH_OVERHEAD_NEXT * (nb - 1)
    return t


def partition_tiles(grid, imgsz, world, batch=256):
    """Deterministic cost-balanced tile -> rank assignment (every rank computes the same one).

    Tiles are grouped by shape (one kernel-launch sequence per shape), shapes ordered by letterboxed area (largest
    first), and the concatenated list is cut into `world` contiguous runs of (nearly) equal ESTIMATED TIME, where the
    estimate charges every batch a fixed overhead on top of its pixels: the rank that inherits the three small ragged
    classes of a grid pays three extra launch sequences and therefore gets fewer full tiles.  The cut is the greedy
    packing for the smallest feasible time bound (bisection), so a rank still sees at most a couple of shape classes.
    Returns per rank: list of ((th, tw), [tile ids])."""
    classes = {}
    for tid, (x0, x1, y0, y1) in enumerate(grid):
        classes.setdefault((y1 - y0, x1 - x0), []).append(tid)
    area = {}
    for (th, tw) in classes:
        lb = L.letterbox(th, tw, imgsz)
        area[(th, tw)] = lb.H * lb.W
    amax = float(max(area.values())) if area else 1.0
    cost = {k: v / amax for k, v in area.items()}
    order = sorted(classes, key=lambda k: (-area[k], -k[0], -k[1]))
    seq = [(shp, t) for shp in order for t in classes[shp]]

    def pack(bound):
        """Greedy contiguous packing: a rank takes tiles while its estimated time stays <= bound."""
        out, cur, segs = [], [], []
        for shp, t in seq:
            trial = list(segs)
            if trial and trial[-1][0] == shp:
                trial[-1] = (shp, trial[-1][1] + 1)
            else:
                trial.append((shp, 1))
            if cur and _rank_cost(trial, cost, batch) > bound and len(out) < world - 1:
                out.append(cur)
                cur, segs = [], []
                trial = [(shp, 1)]
            segs = trial
            if not cur or cur[-1][0] != shp:
                cur.append((shp, []))
            cur[-1][1].append(t)
        out.append(cur)
        while len(out) < world:
            out.append([])
        return out

    def worst(parts):
        return max(_rank_cost([(shp, len(t)) for shp, t in p], cost, batch) for p in parts)
    if world <= 1 or not seq:
        return pack(float("inf"))[:1] + [[] for _ in range(max(world - 1, 0))]
    total = _rank_cost([(shp, len(classes[shp])) for shp in order], cost, batch)
    lo, hi = total / world * 0.5, total + BATCH_OVERHEAD_TILES * world
    for _ in range(40):                                     # bisection on the time bound; pack() is monotone in it
        mid = 0.5 * (lo + hi)
        if worst(pack(mid)) <= mid:
            hi = mid
        else:
            lo = mid
    return pack(hi)


class MosaicSource(object):
    """The image on the HOST (numpy array or memmap of the FITS payload, file byte order) from which a rank uploads only
    the regions its tiles touch: SURVEY.md section 8(e) "each GPU holds ... only its row-band of the mosaic (+ overlap
    halo)"; the reference reads only the tile, caesar_yolo/utils.py:368-394.  A region already resident is reused."""

    def __init__(self, host, big_endian=None):
        self.host = host
        self.ny, self.nx = host.shape
        self.big_endian = (host.dtype.byteorder == ">") if big_endian is None else bool(big_endian)
        self.regions = []                       # (x0, x1, y0, y1, device tensor)
        self.bytes_uploaded = 0

    def region(self, det, x0, x1, y0, y1):
        """-> (device tensor [y1-y0 (or more), ...], origin x, origin y) covering [x0,x1) x [y0,y1)."""
        for a0, a1, b0, b1, t in self.regions:
            if a0 <= x0 and x1 <= a1 and b0 <= y0 and y1 <= b1:
                return t, a0, b0
        file_rows = None
        if (isinstance(self.host, np.memmap) and self.host.dtype.itemsize == 4 and getattr(self.host, "filename", None)
                and (x1 - x0) * 10 >= 9 * self.nx):
            # (nearly) full-width band of a memory-mapped file: take whole rows, which are one contiguous run of the file, and
            # read them with pread() straight into the upload's pinned buffers (faulting the map in page by page took as
            # long as the transfer itself)
            x0, x1 = 0, self.nx
            file_rows = (str(self.host.filename), int(self.host.offset) + y0 * self.nx * 4)
        arr = self.host[y0:y1, x0:x1]                       # a view: the upload copies it chunk by chunk
        t = det.mosaic_to_device(arr, big_endian=self.big_endian, file_rows=file_rows)
        self.regions.append((x0, x1, y0, y1, t))
        self.bytes_uploaded += arr.size * 4
        return t, x0, y0


class TileEngine(object):
    """Runs the per-tile path for this rank's share of a tile grid and merges all ranks' detections.

    `mosaic_dev` is either the whole image resident on the device, or a MosaicSource: then only the bounding box of each
    shape-class segment of THIS rank's tiles is uploaded (the partition cuts the tile list into contiguous runs, so a
    rank's full-size tiles are a band of rows) and tile origins are rebased to it."""

    def __init__(self, detector, mosaic_dev, grid, pre_cfg, imgsz, conf, iou, soft, hard, rank=0, world=1, batch=64):
        self.det, self.mosaic, self.grid = detector, mosaic_dev, [tuple(int(v) for v in t) for t in grid]
        self.pre_cfg, self.imgsz = pre_cfg, int(imgsz)
        self.conf, self.iou, self.soft, self.hard = float(conf), float(iou), float(soft), float(hard)
        self.rank, self.world, self.batch = rank, world, min(int(batch), detector.max_batch)
        parts = partition_tiles(self.grid, self.imgsz, world, self.batch)
        self._parts, self._perm = parts, None
        self.counts = [sum(len(t) for _, t in p) for p in parts]
        self.cap_tiles = max(max(self.counts), 1)
        self.my = parts[rank]
        n_my = self.counts[rank]
        # launch plan, built once: (th, tw, B, origins, first row, device image the origins refer to)
        self.plan, row, self._origins = [], 0, {}
        for (th, tw), tids in self.my:
            img, ox, oy = self.mosaic, 0, 0
            if isinstance(self.mosaic, MosaicSource) and tids:
                g = [self.grid[t] for t in tids]
                img, ox, oy = self.mosaic.region(detector, min(t[0] for t in g), max(t[1] for t in g),
                                                 min(t[2] for t in g), max(t[3] for t in g))
            self._origins[id(img)] = (ox, oy)
            nb = (len(tids) + self.batch - 1) // self.batch        # equal-sized batches (198 tiles -> 50,50,49,49, not 64,64,64,6)
            i = 0
            for k in range(nb):
                n = len(tids) // nb + (1 if k < len(tids) % nb else 0)
                chunk = tids[i:i + n]
                i += n
                xy = [(self.grid[t][0] - ox, self.grid[t][2] - oy) for t in chunk]
                self.plan.append((th, tw, len(chunk), xy, row, img))
                row += len(chunk)
        # Launch order: the small batches (ragged edge classes, < 64 tiles) are slotted between the full ones -- the library runs
        # their forward on a second stream (cy_detect_tiles, "small-batch lane"), where a chain of ~105 tiny kernels costs its
        # share of the chip instead of its latency (4-6 ms each in line on the 16k mosaic).  Output rows stay where they were.
        is_small = lambda p: p[2] < 64 and p[2] * 3 <= detector.max_batch       # the library's rule (cy_detect_tiles)
        big = [p for p in self.plan if not is_small(p)]
        small = [p for p in self.plan if is_small(p)]
        if big and small:
            order = []
            for i, p in enumerate(big):
                order.append(p)
                if i < len(small):
                    order.append(small[i])
            order += small[len(big):]
            self.plan = order
        dev = detector.tdev
        # fixed-capacity record buffer: [tile][300*6 floats | count | status | tile id]; one extra row at the end carries this
        # rank's event counters (degenerate boxes dropped, tiles whose candidates overflowed), so ONE collective moves all
        self.rec = torch.zeros((self.cap_tiles + 1, L.CY_MAX_DET * 6 + 3), dtype=torch.float32, device=dev)
        # per-tile outputs of this rank, in processing order (each batch writes its own slice: batches overlap)
        self.det_all = torch.zeros((max(n_my, 1), L.CY_MAX_DET, 6), dtype=torch.float32, device=dev)
        self.cnt_all = torch.zeros((max(n_my, 1),), dtype=torch.int32, device=dev)
        self.st_all = torch.zeros((max(n_my, 1),), dtype=torch.int32, device=dev)
        order = [t for _, tids in self.my for t in tids]
        self.tid_all = torch.tensor(order if order else [0], dtype=torch.float32, device=dev)
        self.n_my = n_my
        self.gathered = None

    def origin_of(self, img):
        """(x, y) of the mosaic pixel at [0, 0] of a device image of the launch plan (regions are rebased)."""
        return self._origins.get(id(img), (0, 0))

    def run_local(self):
        """Enqueue every batch of this rank's tiles (software-pipelined inside the library); results stay on device."""
        for th, tw, B, xy, row, img in self.plan:
            out = (self.det_all[row:row + B], self.cnt_all[row:row + B], self.st_all[row:row + B])
            self.det.detect_tiles(img, xy, th, tw, self.imgsz, self.pre_cfg, self.conf, self.iou,
                                  self.soft, self.hard, out=out, flush=False)
        if hasattr(self.det, "flush"):
            self.det.flush()
        n = self.n_my
        if n:
            self.rec[:n, :L.CY_MAX_DET * 6] = self.det_all[:n].reshape(n, -1)
            self.rec[:n, -3] = self.cnt_all[:n].float()
            self.rec[:n, -2] = self.st_all[:n].float()
            self.rec[:n, -1] = self.tid_all[:n]
        if hasattr(self.det, "counters"):
            c = self.det.counters(reset=True)
            self.rec[self.cap_tiles, :2] = torch.tensor([c["degenerate_boxes"], c["cand_overflow_tiles"]], dtype=torch.float32).to(self.rec.device)
        return n

    def gather(self):
        """ONE collective: all-gather of the fixed-capacity record buffers (RCCL over xGMI when world > 1).  The records are complete
        on the caller's stream before it is issued: run_local() ends with flush() (the stream waits for the post-processing events of
        the last batches) and fills `rec` with ordinary stream-ordered copies; the process group orders its own stream behind ours.
        With a process group of ONE rank (bench.py CY_BENCH_FORCE_DIST=1, tests/test_gpu_multirank.py) the collective still runs."""
        import torch.distributed as dist
        if self.world > 1 or (dist.is_available() and dist.is_initialized()):
            out = torch.empty((self.world,) + tuple(self.rec.shape), dtype=self.rec.dtype, device=self.rec.device)
            if dist.get_backend() == "gloo":          # CPU rehearsal of the same collective (tests, 1-GPU boxes)
                mine = self.rec.cpu()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine)
                out = torch.stack(parts, 0).to(self.rec.device)
            else:
                dist.all_gather_into_tensor(out, self.rec)
            self.gathered = out
        else:
            self.gathered = self.rec.unsqueeze(0)
        return self.gathered

    def merged_records(self):
        """All ranks' records -> tile-id order -> edge flags + cross-tile merge.  Returns (records [M,8] float64 =
        x1,y1,x2,y2,score,class_id,edge,merged in catalog order, stats).  Valid rows are compacted on device first, so
        only the detections themselves cross PCIe."""
        g = self.gathered
        if getattr(self, "_perm", None) is None:
            # the partition is deterministic: which (rank, row) holds which tile is known on every rank without looking at
            # the data, so "valid rows in tile-id order" is ONE precomputed gather index
            where = {}
            for r, part in enumerate(self._parts):
                row = 0
                for _, tids in part:
                    for t in tids:
                        where[t] = r * g.shape[1] + row
                        row += 1
            self._perm = torch.tensor([where[t] for t in sorted(where)], dtype=torch.int64, device=g.device)
            self._tid_sorted = np.array(sorted(where), np.int32)
        import time as _t
        t0 = _t.time()
        T = int(self._perm.shape[0])
        if hasattr(self.det, "compact_records"):
            if getattr(self, "_hdr", None) is None or self._hdr.shape[0] != 3 * T + 1:
                self._hdr = torch.empty((3 * T + 1,), dtype=torch.int32, device=g.device)
                self._cdet = torch.empty((T * L.CY_MAX_DET * 6,), dtype=torch.float32, device=g.device)
                self._hdr_h = torch.empty((3 * T + 1,), dtype=torch.int32, pin_memory=True)
                self._cdet_h = torch.empty((T * L.CY_MAX_DET * 6,), dtype=torch.float32, pin_memory=True)
            # two launches on the device: per-tile counts / status / prefix, then the valid detections in tile-id order; two small D2H copies
            self.det.compact_records(g, self._perm, self._hdr, self._cdet)
            self._hdr_h.copy_(self._hdr, non_blocking=True)
            torch.cuda.current_stream(g.device).synchronize()
            hdr = self._hdr_h.numpy()
            cnt_h, status_h, total = hdr[:T].astype(np.int64), hdr[T:2 * T].astype(np.int64), int(hdr[3 * T])
            self._cdet_h[:total * 6].copy_(self._cdet[:total * 6], non_blocking=True)
            torch.cuda.current_stream(g.device).synchronize()
            det_h = self._cdet_h[:total * 6].numpy().reshape(total, 6).copy()
        else:                                               # (a detector object without the entry point: host-side tests with a stub)
            rows = g.reshape(-1, g.shape[-1]).index_select(0, self._perm)          # [T, 1803] in tile-id order
            meta = rows[:, -3:-1].cpu().numpy()
            status_h = meta[:, 1].astype(np.int64)
            cnt_h = np.where(status_h == 0, meta[:, 0].astype(np.int64), 0)
            keep = torch.arange(L.CY_MAX_DET, device=g.device)[None, :] < torch.from_numpy(cnt_h).to(g.device)[:, None]
            det_h = np.ascontiguousarray(rows[:, :L.CY_MAX_DET * 6].reshape(-1, L.CY_MAX_DET, 6)[keep].cpu().numpy(), np.float32)
        stats = {"tiles": T, "skipped": int((status_h != 0).sum()), "per_tile_detections": int(cnt_h.sum())}
        ev = g[:, self.cap_tiles, :2].sum(0).cpu().numpy()                         # the ranks' event counters (last row)
        stats["degenerate_boxes"], stats["cand_overflow_tiles"] = int(ev[0]), int(ev[1])
        if ev[0] or ev[1]:
            logger.warning("%d degenerate boxes dropped before the IoU merge (the reference aborts on them: utils.py:78-81); "
                           "%d tiles exceeded the candidate capacity (raise max_cand)" % (int(ev[0]), int(ev[1])))
        dtile_h = np.ascontiguousarray(np.repeat(self._tid_sorted, cnt_h))
        t1 = _t.time()
        if getattr(self, "_tiles_np", None) is None:              # the grid as the C-ABI wants it, converted once
            self._tiles_np = np.ascontiguousarray(np.array(self.grid, np.int32).reshape(-1, 4))
        out = merge_records(det_h, dtile_h, self._tiles_np)
        stats["merge_host_ms"] = 1e3 * (_t.time() - t1)
        stats["merge_d2h_ms"] = 1e3 * (t1 - t0)
        return out, stats

    def tile_results(self, names, image_id):
        """Per-tile `Analyzer.results` dicts of every tile that was not rejected, keyed by tile id: what each TileTask's
        Analyzer holds after predict (caesar_yolo/evaluation.py:418-469 with obj_name_tag "t<tid>" and the tile origin,
        caesar_yolo/inference.py:107, :230-232).  Needs gather() first; rank 0 writes the per-tile files from it."""
        g = self.gathered
        rows = g[:, :self.cap_tiles].reshape(-1, g.shape[-1]).cpu().numpy()
        out = {}
        for r, part in enumerate(self._parts):
            row = 0
            for _, tids in part:
                for t in tids:
                    rec = rows[r * self.cap_tiles + row]
                    row += 1
                    if int(rec[-2]) != 0:
                        continue
                    x0, x1, y0, y1 = self.grid[t]
                    dd = rec[:L.CY_MAX_DET * 6].reshape(L.CY_MAX_DET, 6)[:int(rec[-3])]
                    out[t] = {"image_id": image_id, "objs": objs_from_detections(dd, names, x1 - x0, y1 - y0, x0, y0, "t%d" % t)}
        return out

    def catalog(self, names):
        """-> (the reference's list of source dicts, stats)."""
        rec, stats = self.merged_records()
        return records_to_sources(rec, names), stats


def merge_records(det, dtile, grid):
    """det [n,6] float32 tile-pixel detections grouped by ascending tile id, dtile [n] int32 -> merged records [M,8]."""
    n = det.shape[0]
    if n == 0:
        return np.zeros((0, 8), np.float64)
    tiles = grid if isinstance(grid, np.ndarray) else np.ascontiguousarray(np.array(grid, np.int32).reshape(-1, 4))
    T = tiles.shape[0]
    lib = L.load()
    ip, fp, dp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    rec = np.empty((n, 8), np.float64)                  # every row is written by cy_make_tile_records
    L.check(lib.cy_make_tile_records(det.ctypes.data_as(fp), dtile.ctypes.data_as(ip), n, tiles.ctypes.data_as(ip), T,
                                     rec.ctypes.data_as(dp)))
    out = np.empty((n, 8), np.float64)                  # rows [0, m) are written, the rest is dropped
    m = L.check(lib.cy_merge_edge_sources(rec.ctypes.data_as(dp), n, tiles.ctypes.data_as(ip), T, out.ctypes.data_as(dp)))
    return out[:m]


def build_catalog(rows, grid, names):
    """rows: [ntiles, 300*6+3] host records (any order).  Returns (sources list, per-tile stats dict)."""
    tid = rows[:, -1].astype(np.int64)
    rows = rows[np.argsort(tid, kind="stable")]
    cnt = rows[:, -3].astype(np.int64)
    status = rows[:, -2].astype(np.int64)
    cnt = np.where(status == 0, cnt, 0)
    stats = {"tiles": int(rows.shape[0]), "skipped": int(np.sum(status != 0)), "per_tile_detections": int(cnt.sum())}
    keep = np.arange(L.CY_MAX_DET)[None, :] < cnt[:, None]
    det = np.ascontiguousarray(rows[:, :L.CY_MAX_DET * 6].reshape(-1, L.CY_MAX_DET, 6)[keep], np.float32)
    dtile = np.ascontiguousarray(np.repeat(rows[:, -1].astype(np.int32), cnt))
    return records_to_sources(merge_records(det, dtile, grid), names), stats


_EDGE = {0.0: 0, 1.0: 1, 2.0: True}        # the reference mixes int and bool in this field


def records_to_sources(out, names):
    src = []
    for i in range(out.shape[0]):
        x1, y1, x2, y2, score, cid, e, m = out[i]
        src.append({"name": "S%d" % (i + 1), "x1": float(x1), "x2": float(x2), "y1": float(y1), "y2": float(y2),
                    "class_id": int(cid), "class_name": str(names[int(cid)]), "score": float(score),
                    "edge": _EDGE[float(e)], "merged": bool(m)})
    return src


def objs_from_detections(det, names, nx, ny, xmin=0, ymin=0, tag=""):
    """Analyzer.make_json_results (caesar_yolo/evaluation.py:418-469) for one frame: int() truncation, edge rule."""
    objs = []
    for i in range(det.shape[0]):
        x1, y1, x2, y2 = (int(v) for v in det[i, :4])
        at_edge = (x1 <= 0 or x1 >= nx - 1 or x2 <= 0 or x2 >= nx - 1 or y1 <= 0 or y1 >= ny - 1 or y2 <= 0 or y2 >= ny - 1)
        cid = int(det[i, 5])
        objs.append({"name": "S%d" % (i + 1) + ("_" + tag if tag else ""), "x1": float(xmin + x1), "x2": float(xmin + x2),
                     "y1": float(ymin + y1), "y2": float(ymin + y2), "class_id": cid, "class_name": str(names[cid]),
                     "score": float(det[i, 4]), "edge": int(at_edge)})
    return objs


class SFinder(object):
    """caesar_yolo/inference.py:280: `SFinder(model, config)`; `run()` / `run_parallel()` return 0 or -1."""

    def __init__(self, model, config):
        self.model, self.config = model, config
        self.sources = {"sources": []}
        self.results = {}
        self.image_id = ""
        self.nx = self.ny = -1
        self.outfile_json = config.get('outfile_json', '')
        self.write_to_json = config.get('save_catalog', True)
        self.write_to_ds9 = config.get('save_region', True)      # caesar_yolo/inference.py:331
        self.outfile_ds9 = ""
        self.runtime = 0.0
        self.stats = {}

    # ---- shared
    def _load_mosaic(self, resident=True):
        """resident=True: the whole image on the device (single frame).  False: a MosaicSource over the memory-mapped
        FITS payload; the tile engine then uploads only this rank's regions."""
        path = self.config['image_path']
        if os.path.splitext(path)[1] != '.fits':
            logger.error("Only FITS images are supported on the HIP path")
            return None
        res = utils.read_fits_image(path)
        if res is None:
            return None
        data, self.header = res
        self.image_id = utils.image_id_of(path)
        self.ny, self.nx = data.shape
        self._beam_info()
        det = self.model.engine(self._device())
        if not resident:
            return det, MosaicSource(data, big_endian=True)
        src = MosaicSource(data, big_endian=True)          # whole image = one full-width band: the pread() upload path
        return det, src.region(det, 0, self.nx, 0, self.ny)[0]

    def _beam_info(self):
        """Beam / pixel metadata of the reference's read_img (caesar_yolo/inference.py:430-468): beamArea = pi*BMAJ*BMIN /
        (4 ln 2) / |CDELT1*CDELT2| pixels, 0 when any of the five keywords is missing (a warning per keyword, as there)."""
        h = self.header
        self.wcs = wcs_of_header(h)                       # caesar_yolo/inference.py:473
        self.beamArea, ok = 0, True
        for key, attr in (("CDELT1", "dX"), ("CDELT2", "dY"), ("BMAJ", "bmaj"), ("BMIN", "bmin"), ("BPA", "pa")):
            if key not in h:
                logger.warning("%s keyword missing in header!" % key)
                ok = False
            else:
                setattr(self, attr, h[key])
        if ok:
            self.pixelArea = np.abs(self.dX * self.dY)
            self.beamArea = np.pi * self.bmaj * self.bmin / (4 * np.log(2)) / self.pixelArea
            logger.info("Image info: beam(%f,%f,%f), beamArea=%f, dx=%f, dy=%f, pixArea=%g" % (
                self.bmaj * 3600, self.bmin * 3600, self.pa, self.beamArea, self.dX * 3600, self.dY * 3600, self.pixelArea))

    def _device(self):
        devs = self.config.get('devices', ['0'])
        rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.config.get('use_multi_gpu') and len(devs) > rank:
            return devs[rank]                      # caesar_yolo/inference.py:207-210: device = devices[procId]
        d = devs[0]
        return int(os.environ.get("LOCAL_RANK", "0")) if d == "" else d      # 'cpu': model._dev_index picks this rank's GPU and warns

    def _pre_cfg(self):
        dp = self.config.get('preprocess_fcn')
        return dp.program() if dp is not None else no_preprocessing()

    def _thr(self):
        c = self.config
        return c['score_thr'], c['iou_thr'], c['merge_overlap_iou_thr_soft'], c['merge_overlap_iou_thr_hard']

    # ---- serial: one frame (reference :485-552: read the image, hand it to an Analyzer)
    def run(self):
        from .evaluation import Analyzer
        path = self.config['image_path']
        if os.path.splitext(path)[1] != '.fits':
            logger.error("Only FITS images are supported on the HIP path")
            return -1
        res = utils.read_fits_image(path)
        if res is None:
            logger.error("Failed to read image %s!" % path)
            return -1
        data, self.header = res
        # sub-image of the serial run (caesar_yolo/inference.py:499-505 -> utils.read_fits_crop, utils.py:340-394): ranges all 0 / -1 =
        # the whole image; otherwise [ymin:ymax, xmin:xmax] with the max pixel EXCLUDED, the same argument errors, and catalog
        # coordinates relative to the crop (the reference hands the crop to Analyzer.predict without an origin)
        c = self.config
        rng = [int(c.get(k, 0)) for k in ('image_xmin', 'image_xmax', 'image_ymin', 'image_ymax')]
        if not all(v in (0, -1) for v in rng):
            ixmin, ixmax, iymin, iymax = rng
            if ixmin < 0 or ixmax < 0 or iymin < 0 or iymax < 0:
                logger.error("ixmin/ixmax/iymin/iymax must be >0")
                return -1
            if ixmax <= ixmin or iymax <= iymin:
                logger.error("ixmax must be >ixmin and iymax >iymin!")
                return -1
            data = data[iymin:iymax, ixmin:ixmax]                  # a view of the memory map: only the crop is uploaded
            if data.size == 0:
                logger.error("Failed to read data in range[%d:%d,%d:%d] from file %s!" % (iymin, iymax, ixmin, ixmax, path))
                return -1
        self.image_id = utils.image_id_of(path)
        self.ny, self.nx = data.shape
        self._beam_info()
        analyzer = Analyzer(self.model, self.config)
        analyzer.device = self._device()
        analyzer.outfile_json = self.outfile_json
        if analyzer.predict(image=data, image_id=self.image_id, header=self.header) < 0:
            logger.error("Failed to run model prediction on image %s!" % path)
            return -1
        self.results = analyzer.results
        if not analyzer.bboxes_final:
            logger.info("No object detected in image %s ..." % path)
        else:
            logger.info("#%d objects found in image %s ..." % (len(analyzer.bboxes_final), path))
        return 0

    def _write_tile_outputs(self, eng):
        """--save_tile_catalog / --save_tile_region (caesar_yolo/inference.py:330-350, :220-227, :1020-1021): one
        catalog_<id>_tid<N>.json / .reg per tile that ran.  The reference writes them from each worker's Analyzer; here rank 0
        writes the same files after the gather (Analyzer's colour map for the regions)."""
        c = self.config
        sj, sr = c.get('save_tile_catalog', False), c.get('save_tile_region', False)
        if not (sj or sr):
            return
        res = eng.tile_results(self.model.names, self.image_id)
        for tid, r in sorted(res.items()):
            stem = str(self.image_id) + '_tid' + str(tid)
            if sj:
                with open('catalog_' + stem + '.json', 'w') as fp:
                    json.dump(r, fp, indent=2, sort_keys=True)
            if sr:
                utils.write_ds9_regions('catalog_' + stem + '.reg', r["objs"], color_map=utils.CLASS_COLOR_MAP_DS9_FRAME,
                                        merged_tag=False)

    def _write_tile_images(self, eng):
        """--save_tile_img (:228-229 -> Analyzer.write_fits, evaluation.py:550-554): timg_<id>_tid<N>.fits = channel 0 of the
        PREPROCESSED tile, float64.  Every rank writes its own tiles (it holds their pixels), like the reference's workers;
        tiles the pipeline rejected write nothing."""
        if not self.config.get('save_tile_img', False):
            return
        grid_id = {(g[0], g[2], g[3] - g[2], g[1] - g[0]): t for t, g in enumerate(eng.grid)}
        for th, tw, B, xy, row, img in eng.plan:
            planes, status = eng.det.preproc_planes(img, xy, th, tw, eng.pre_cfg)
            ch0, status = planes[:, 0].cpu().numpy(), status.cpu().numpy()
            ox, oy = eng.origin_of(img)
            for b, (x, y) in enumerate(xy):
                if status[b] == 0:
                    tid = grid_id[(x + ox, y + oy, th, tw)]
                    utils.write_fits_image('timg_' + str(self.image_id) + '_tid' + str(tid) + '.fits', ch0[b], bitpix=-64)

    # ---- tiled (reference :578-658)
    def run_parallel(self):
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        t0 = time.time()
        m = self._load_mosaic(resident=False)
        if m is None:
            return -1
        det, mosaic = m
        c = self.config
        tsx, tsy = (c['tile_xsize'], c['tile_ysize']) if c['split_image_in_tiles'] else (self.nx, self.ny)
        stx, sty = (c['tile_xstep'], c['tile_ystep']) if c['split_image_in_tiles'] else (1, 1)
        grid = utils.generate_tiles(0, self.nx - 1, 0, self.ny - 1, tsx, tsy, stx, sty)
        if grid is None:
            logger.warning("[PROC %d] Failure in create tile tasks, exit..." % rank)
            return -1
        conf, iou, soft, hard = self._thr()
        eng = TileEngine(det, mosaic, grid, self._pre_cfg(), c['img_size'], conf, iou, soft, hard, rank, world,
                         c.get('tile_batch', 64))
        # caesar_yolo/inference.py:1151-1160: the reference refuses a run in which a worker holds more than
        # max_ntasks_per_worker tiles (its default, 100, would refuse BASELINE configs 3-5 at any rank count up to 16).  The
        # batched engine has no such limit, so the guard applies only when the option was given explicitly (run.py passes
        # None otherwise); then it behaves like the reference: warning, -1.
        cap = c.get('max_ntasks_per_worker')
        if cap is not None and max(eng.counts) > cap:
            logger.warning("[PROC %d] Too many tasks per worker exceeded (thr=%d)!" % (rank, cap))
            return -1
        eng.run_local()
        eng.gather()
        self._write_tile_images(eng)
        if rank == 0:
            self._write_tile_outputs(eng)
            src, self.stats = eng.catalog(self.model.names)
            self.sources = {"sources": src}
            if self.write_to_json:
                out = self.outfile_json or ('catalog_' + str(self.image_id) + '.json')
                with open(out, 'w') as fp:
                    json.dump(self.sources, fp, indent=2, sort_keys=True)
            if self.write_to_ds9:                                 # caesar_yolo/inference.py:1188-1194
                utils.write_ds9_regions(self.outfile_ds9 or ('ds9_' + str(self.image_id) + '.reg'), src)
        if world > 1:
            dist.barrier()
        self.runtime = time.time() - t0
        if rank == 0:
            logger.info("[PROC %d] Run completed in %d seconds" % (rank, self.runtime))
        return 0
