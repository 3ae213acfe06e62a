"""`YOLO`-compatible detector object backed by the HIP library.

Mirrors the slice of `ultralytics.YOLO` that the reference touches:
  * construction with a weights path            scripts/run.py:347
  * `.names` (dict class id -> label)           caesar_yolo/evaluation.py:46-47, :265
  * `model(image, device=, imgsz=, conf=, iou=, **ignored)` returning an iterable of results whose
    `.boxes.xyxy / .conf / .cls` support `.cpu().numpy()`      caesar_yolo/evaluation.py:181-193, :261-265
plus the additive batched entry the tile scheduler uses (`detect_tiles`): B same-shape tiles cropped from an
HBM-resident mosaic -> merged detections, everything on device.

PyTorch is used only for device memory and streams; all arithmetic is in libcaesar_yolo_hip.so.  There is no CPU
path: constructing a detector without a visible GPU, or without the built library, raises.
"""
import ctypes as C
import logging
import os
import numpy as np
import torch

from . import lib as L
from . import weights as W


_CPU_WARNED = False


def _dev_index(device):
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("", "cuda", "gpu"):
        return torch.cuda.current_device()
    if s.startswith("cuda:"):
        return int(s.split(":")[1])
    if s.isdigit():
        return int(s)
    if s == "cpu":
        # The reference's DEFAULT is --devices=cpu (scripts/run.py:130) and Analyzer passes it straight to the model call
        # (caesar_yolo/evaluation.py:183).  This build has no CPU path; raising here would make every tile fail inside the
        # reference's try/except (evaluation.py:194-196) and the run would "succeed" with an empty catalog.  So 'cpu' selects
        # the process's GPU (LOCAL_RANK under torch.distributed, else 0) -- the same rule scripts/run.py applies -- and says so once.
        global _CPU_WARNED
        if not _CPU_WARNED:
            logging.getLogger("caesar_yolo_amd").warning(
                "device='cpu' requested: this build has no CPU path, running on GPU %s instead", os.environ.get("LOCAL_RANK", "0"))
            _CPU_WARNED = True
        return int(os.environ.get("LOCAL_RANK", "0"))
    raise L.CyError("unknown device %r" % (device,))


class HipDetector(object):
    """One library context on one GPU.  Thin, explicit wrappers over the C-ABI stage entry points."""

    def __init__(self, weights_path, device=0, precision="fp16x3", max_batch=64, max_imgsz=640, max_cand=0):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.CyError("no GPU visible: the HIP detector cannot run (no CPU fallback exists)")
        self.device = _dev_index(device)
        self.tdev = torch.device("cuda", self.device)
        if isinstance(precision, str):
            if precision.lower() not in L.PRECISIONS:
                raise L.CyError("unknown precision %r (fp16 | fp16x3 | fp32)" % (precision,))
            precision = L.PRECISIONS[precision.lower()]
        if precision not in (L.F16, L.F32, L.F16X3):
            raise L.CyError("unknown precision %r" % (precision,))
        self.precision = precision
        # dtype of the tensors that cross the C-ABI (network input, conv test entry): fp16 in the fp16 context, fp32 otherwise
        self.dtype = torch.float16 if self.precision == L.F16 else torch.float32
        self.max_batch = int(max_batch)
        m = (int(max_imgsz) + 31) // 32 * 32
        cfg = L.cy_config(self.precision, self.max_batch, m, m, int(max_cand))
        self.ctx = C.c_void_p()
        L.check(self.lib.cy_create(self.device, C.byref(cfg), C.byref(self.ctx)))
        L.check(self.lib.cy_load_weights(self.ctx, os.fsencode(weights_path)), self.ctx)
        nc = self.lib.cy_num_classes(self.ctx)
        self.names = {i: self.lib.cy_class_name(self.ctx, i).decode() for i in range(nc)}
        self.nc = nc
        # pinned staging buffers of the mosaic upload: part of the context (the first pinned allocation of a process costs
        # ~110 ms of runtime initialisation, which belongs here and not in the first image's ingest)
        self._stage = [torch.empty((self.STAGE_BYTES // 4,), dtype=torch.int32, pin_memory=True) for _ in range(2)]

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.cy_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def _chk(self, rc):
        return L.check(rc, self.ctx)

    # ---- stages
    def mosaic_to_device(self, data, big_endian=None, file_rows=None):
        """Host fp32 image [H,W] (either byte order) -> resident device mosaic with read_fits value semantics.
        file_rows = (path, byte offset of the first row of `data` in that file): `data` is a run of whole rows of a memory-mapped
        file; a large one is then read with pread() straight into the pinned staging buffers (no page faults on the map)."""
        arr = np.asarray(data)
        if arr.dtype.kind != "f" or arr.dtype.itemsize != 4:
            arr = arr.astype("<f4")
        if big_endian is None:
            big_endian = arr.dtype.byteorder == ">"
        if arr.ndim == 2 and arr.nbytes >= self.STAGED_UPLOAD_MIN_BYTES:
            t = self._staged_upload(arr, file_rows)
        else:
            raw = np.ascontiguousarray(arr).view(np.uint32).view(np.int32)
            t = torch.from_numpy(raw.copy() if not raw.flags.writeable else raw).to(self.tdev).view(torch.float32)
        self._chk(self.lib.cy_mosaic_prepare(self.ctx, self._p(t), t.numel(), int(bool(big_endian)), self._stream()))
        return t

    STAGED_UPLOAD_MIN_BYTES = 64 << 20
    STAGE_BYTES = 32 << 20
    STAGE_THREADS = 4

    def _staged_upload(self, arr, file_rows=None):
        """Large host image (typically a slice of the memory-mapped FITS payload, file byte order) -> device, in row chunks
        through two pinned staging buffers: filling chunk i+1 (several threads: pread() of the file when the rows are whole
        rows of it, else numpy copies out of the map; both release the GIL) runs while chunk i is on the wire."""
        from concurrent.futures import ThreadPoolExecutor
        H, Wd = arr.shape
        fd = os.open(file_rows[0], os.O_RDONLY) if file_rows else -1
        out = torch.empty((H, Wd), dtype=torch.int32, device=self.tdev)
        rows = max(1, min(H, self.STAGE_BYTES // (Wd * 4)))
        if getattr(self, "_stage", None) is None or self._stage[0].numel() < rows * Wd:         # (a single row wider than a buffer)
            self._stage = [torch.empty((rows * Wd,), dtype=torch.int32, pin_memory=True) for _ in range(2)]
        views = [st[:rows * Wd].view(rows, Wd).numpy().view(arr.dtype) for st in self._stage]      # same item size: a raw byte copy
        done = [None, None]
        nthr = self.STAGE_THREADS
        with ThreadPoolExecutor(nthr) as pool:
            for i, y in enumerate(range(0, H, rows)):
                k = i & 1
                if done[k] is not None:
                    done[k].synchronize()                        # the previous transfer out of this buffer has finished
                n = min(rows, H - y)
                cuts = [n * j // nthr for j in range(nthr + 1)]
                if fd >= 0:
                    def fill(j, k=k, y=y, cuts=cuts):
                        if cuts[j + 1] > cuts[j]:
                            buf = memoryview(views[k][cuts[j]:cuts[j + 1]]).cast("B")
                            got = os.preadv(fd, [buf], file_rows[1] + (y + cuts[j]) * Wd * 4)
                            if got != len(buf):
                                raise L.CyError("short read of %s" % (file_rows[0],))
                else:
                    def fill(j, k=k, y=y, cuts=cuts):
                        np.copyto(views[k][cuts[j]:cuts[j + 1]], arr[y + cuts[j]:y + cuts[j + 1]])
                list(pool.map(fill, range(nthr)))
                out[y:y + n].copy_(self._stage[k][:n * Wd].view(n, Wd), non_blocking=True)
                done[k] = torch.cuda.Event()
                done[k].record()
        for e in done:
            if e is not None:
                e.synchronize()
        if fd >= 0:
            os.close(fd)
        return out.view(torch.float32)

    def preproc(self, mosaic, tiles_xy, th, tw, imgsz, cfg):
        B = len(tiles_xy)
        lb = L.letterbox(th, tw, imgsz)
        netin = torch.empty((B, lb.H, lb.W, 4), dtype=self.dtype, device=self.tdev)
        status = torch.empty((B,), dtype=torch.int32, device=self.tdev)
        t = (C.c_int * (2 * B))(*[int(v) for xy in tiles_xy for v in xy])
        self._chk(self.lib.cy_preproc(self.ctx, self._p(mosaic), mosaic.shape[0], mosaic.shape[1], t, B, th, tw, imgsz,
                                      C.byref(cfg), self._p(netin), self._p(status), self._stream()))
        return netin, status, lb

    def preproc_planes(self, mosaic, tiles_xy, th, tw, cfg):
        """-> (float64 [B,3,th,tw] preprocessed images in image channel order, status [B]) on device."""
        B = len(tiles_xy)
        planes = torch.empty((B, 3, th, tw), dtype=torch.float64, device=self.tdev)
        status = torch.empty((B,), dtype=torch.int32, device=self.tdev)
        t = (C.c_int * (2 * B))(*[int(v) for xy in tiles_xy for v in xy])
        self._chk(self.lib.cy_preproc_planes(self.ctx, self._p(mosaic), mosaic.shape[0], mosaic.shape[1], t, B, th, tw,
                                             C.byref(cfg), self._p(planes), self._p(status), self._stream()))
        return planes, status

    def preproc_params(self, B):
        out = np.zeros((B, 3, L.CY_MAX_STAGES, 4), np.float64)
        self._chk(self.lib.cy_preproc_params(self.ctx, out.ctypes.data_as(C.POINTER(C.c_double)), B))
        return out

    def letterbox_pack(self, planes, imgsz):
        """planes: device float64 [B,3,h0,w0] (image channel order, [0,255])."""
        B, _, h0, w0 = planes.shape
        lb = L.letterbox(h0, w0, imgsz)
        netin = torch.empty((B, lb.H, lb.W, 4), dtype=self.dtype, device=self.tdev)
        self._chk(self.lib.cy_letterbox_pack(self.ctx, self._p(planes), B, h0, w0, imgsz, self._p(netin), self._stream()))
        return netin, lb

    def forward(self, netin):
        B, H, Wd, _ = netin.shape
        A = self.lib.cy_num_anchors(H, Wd)
        pred = torch.empty((B, A, 64 + self.nc), dtype=torch.float32, device=self.tdev)
        self._chk(self.lib.cy_forward(self.ctx, self._p(netin), B, H, Wd, self._p(pred), self._stream()))
        return pred

    def weight_passes(self):
        """fp16x3 context: (convolutions on the two-pass form, on the three-pass form); (0, 0) in the other contexts."""
        out = (C.c_int * 2)()
        self._chk(self.lib.cy_weight_passes(self.ctx, out))
        return int(out[0]), int(out[1])

    def profile(self, on):
        self._chk(self.lib.cy_profile_enable(self.ctx, int(on)))            # True/1: every forward call; N > 1: every N-th

    def profile_summary(self, lane=-1):
        """lane: -1 all launches, 0 the main lane of detect_tiles (full batches), 1 its small-batch lane."""
        ent = (L.cy_prof_entry * 32)()
        n = self._chk(self.lib.cy_profile_summary_lane(self.ctx, ent, 32, int(lane)))
        return [dict(kernel=ent[i].kernel.decode(), ms=ent[i].ms, flops=ent[i].flops, launches=ent[i].launches) for i in range(n)]

    def profile_layers(self):
        ent = (L.cy_prof_entry * 256)()
        n = self._chk(self.lib.cy_profile_layers(self.ctx, ent, 256))
        return [dict(name=ent[i].kernel.decode(), ms=ent[i].ms, flops=ent[i].flops, launches=ent[i].launches) for i in range(n)]

    def read_conv(self, name, shape_hint_elems):
        buf = np.zeros(shape_hint_elems, np.float32)
        dims = (C.c_int * 4)()
        self._chk(self.lib.cy_debug_read_conv(self.ctx, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)),
                                              buf.size, dims))
        d = tuple(dims)
        return buf[:int(np.prod(d))].reshape(d)

    def decode_nms(self, pred, H, Wd, h0, w0, conf, iou):
        B = pred.shape[0]
        det = torch.zeros((B, L.CY_MAX_DET, 6), dtype=torch.float32, device=self.tdev)
        anch = torch.zeros((B, L.CY_MAX_DET), dtype=torch.int32, device=self.tdev)
        cnt = torch.zeros((B,), dtype=torch.int32, device=self.tdev)
        self._chk(self.lib.cy_decode_nms(self.ctx, self._p(pred), B, H, Wd, h0, w0, conf, iou, self._p(det),
                                         self._p(anch), self._p(cnt), self._stream()))
        return det, anch, cnt

    def iou_merge(self, det, cnt, score_thr, soft, hard):
        B = det.shape[0]
        out = torch.zeros_like(det)
        ocnt = torch.zeros((B,), dtype=torch.int32, device=self.tdev)
        osrc = torch.zeros((B, L.CY_MAX_DET), dtype=torch.int32, device=self.tdev)
        self._chk(self.lib.cy_iou_merge(self.ctx, self._p(det), self._p(cnt), B, score_thr, soft, hard, self._p(out),
                                        self._p(ocnt), self._p(osrc), self._stream()))
        return out, ocnt, osrc

    def flush(self):
        """Order the current stream behind every batch queued by detect_tiles (they run on internal side streams)."""
        self._chk(self.lib.cy_detect_flush(self.ctx, self._stream()))

    def compact_records(self, gathered, perm, hdr, out):
        """gathered [R, rows, 1803] fp32, perm [T] int64 (row of tile t over all ranks) -> hdr [3T+1] int32, out [T*300*6] fp32 (device)."""
        T = int(perm.shape[0])
        rows = int(gathered.numel() // gathered.shape[-1])
        self._chk(self.lib.cy_compact_records_ctx(self.ctx, self._p(gathered), rows, self._p(perm), T, int(gathered.shape[-1]), self._p(hdr),
                                                  self._p(out), self._stream()))

    def fence(self):
        """Work queued on the current stream since the last detect_tiles call (an upload into a mosaic buffer already in use, a
        memset of an output buffer) is ordered before the next call's internal streams (cy_detect_fence)."""
        self._chk(self.lib.cy_detect_fence(self.ctx, self._stream()))

    def counters(self, reset=False):
        """-> dict(degenerate_boxes, cand_overflow_tiles) accumulated since the last reset (synchronises the device)."""
        out = (C.c_longlong * 4)()
        self._chk(self.lib.cy_detect_counters(self.ctx, out, int(bool(reset))))
        return {"degenerate_boxes": int(out[0]), "cand_overflow_tiles": int(out[1]), "median_bracket_hits": int(out[2]),
                "median_bracket_misses": int(out[3])}

    def detect_tiles(self, mosaic, tiles_xy, th, tw, imgsz, cfg, conf, iou, soft, hard, out=None, flush=True):
        """Whole per-tile path for B same-shape tiles.  Returns (det [B,300,6], count [B], status [B]) on device.
        With flush=False consecutive calls overlap (give each its own `out` buffers and call flush() before reading)."""
        B = len(tiles_xy)
        if out is None:
            det = torch.empty((B, L.CY_MAX_DET, 6), dtype=torch.float32, device=self.tdev)
            cnt = torch.empty((B,), dtype=torch.int32, device=self.tdev)
            status = torch.empty((B,), dtype=torch.int32, device=self.tdev)
        else:
            det, cnt, status = out
        t = (C.c_int * (2 * B))(*[int(v) for xy in tiles_xy for v in xy])
        self._chk(self.lib.cy_detect_tiles(self.ctx, self._p(mosaic), mosaic.shape[0], mosaic.shape[1], t, B, th, tw,
                                           imgsz, C.byref(cfg), conf, iou, soft, hard, self._p(det), self._p(cnt),
                                           self._p(status), self._stream()))
        if flush:
            self.flush()
        return det, cnt, status

    def bottleneck64(self, x_nhwc, w1, b1, w2, b2, shortcut=True):
        """Fused 64-channel Bottleneck (kernel-level test entry): x [B,H,W,64] fp16 -> [B,H,W,64]."""
        B, H, Wd, Cin = x_nhwc.shape
        out = torch.empty_like(x_nhwc)
        arrs = [np.ascontiguousarray(v, np.float32) for v in (w1, b1, w2, b2)]
        ptr = [v.ctypes.data_as(C.POINTER(C.c_float)) for v in arrs]
        self._chk(self.lib.cy_bottleneck64(self.ctx, self._p(x_nhwc), B, H, Wd, ptr[0], ptr[1], ptr[2], ptr[3], int(bool(shortcut)),
                                           self._p(out), self._stream()))
        return out

    def conv_bn_silu(self, x_nhwc, w, b, k, s, act=True, res=None):
        B, Hi, Wi, Cin = x_nhwc.shape
        Cout = w.shape[0]
        pad = k // 2
        Ho, Wo = (Hi + 2 * pad - k) // s + 1, (Wi + 2 * pad - k) // s + 1
        out = torch.empty((B, Ho, Wo, Cout), dtype=self.dtype, device=self.tdev)
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        self._chk(self.lib.cy_conv_bn_silu(self.ctx, self._p(x_nhwc), B, Hi, Wi, Cin,
                                           w.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                           Cout, k, s, int(act), self._p(res) if res is not None else None,
                                           self._p(out), self._stream()))
        return out


class _Boxes(object):
    def __init__(self, det):
        self.xyxy, self.conf, self.cls = det[:, :4], det[:, 4], det[:, 5]


class Results(object):
    def __init__(self, det):
        self.boxes = _Boxes(det)


class YOLO(object):
    """Drop-in for `ultralytics.YOLO(weights)` on the reference's detect path.

    `weights` is a CYW1 file (caesar_yolo_amd.weights) or an ultralytics YOLOv8 detection `.pt` checkpoint (read as
    data by caesar_yolo_amd.pt_import: no ultralytics needed); the string "seeded:<scale>:<nc>[:seed]" builds the
    deterministic random-init checkpoint used by the tests and the benchmark (no trained weights ship with the
    reference)."""

    def __init__(self, weights, precision="fp16x3", max_batch=64, max_imgsz=640, device=None):
        # precision: "fp16x3" (default: the parity context -- the reference runs ultralytics in fp32), "fp32" (exact, slow), "fp16" (3x faster)
        self._wpath = self._resolve(weights)
        self._kw = dict(precision=precision, max_batch=max_batch, max_imgsz=max_imgsz)
        self._det = None
        self._dev = device
        scale, names, _, _ = W.read_cyw_header(self._wpath)
        self.names = names
        self.scale = scale

    @staticmethod
    def _resolve(weights):
        if isinstance(weights, str) and weights.startswith("seeded:"):
            parts = weights.split(":")
            scale, nc = parts[1], int(parts[2])
            seed = int(parts[3]) if len(parts) > 3 else 20260104
            import tempfile
            cache = os.environ.get("CAESAR_YOLO_CACHE", os.path.join(tempfile.gettempdir(), "caesar_yolo_amd_%d" % os.getuid()))
            os.makedirs(cache, exist_ok=True)
            path = os.path.join(cache, "seeded_%s_nc%d_%d.cyw" % (scale, nc, seed))
            if not os.path.exists(path):
                tmp = path + ".%d.tmp" % os.getpid()
                W.make_seeded_file(tmp, scale, nc, seed)
                os.replace(tmp, path)
            return path
        if isinstance(weights, str) and weights.startswith("seeded11:"):           # seeded YOLO11: "seeded11:<scale>:<nc>[:seed]"
            parts = weights.split(":")
            scale, nc = parts[1], int(parts[2])
            seed = int(parts[3]) if len(parts) > 3 else 11
            import tempfile
            cache = os.environ.get("CAESAR_YOLO_CACHE", os.path.join(tempfile.gettempdir(), "caesar_yolo_amd_%d" % os.getuid()))
            os.makedirs(cache, exist_ok=True)
            path = os.path.join(cache, "seeded11_%s_nc%d_%d.cyw" % (scale, nc, seed))
            if not os.path.exists(path):
                tmp = path + ".%d.tmp" % os.getpid()
                W.make_seeded11_file(tmp, scale, nc, seed)
                os.replace(tmp, path)
            return path
        if not os.path.isfile(weights):
            raise FileNotFoundError(weights)
        with open(weights, "rb") as fp:
            magic = fp.read(4)
        if magic in (b"CYW1", b"CYW2"):
            return weights
        # an ultralytics checkpoint (scripts/run.py:347 passes the .pt path): converted once, without unpickling any of
        # its classes (pt_import), and cached next to the seeded files
        import tempfile
        from . import pt_import
        st = os.stat(weights)
        cache = os.environ.get("CAESAR_YOLO_CACHE", os.path.join(tempfile.gettempdir(), "caesar_yolo_amd_%d" % os.getuid()))
        os.makedirs(cache, exist_ok=True)
        path = os.path.join(cache, "%s_%d_%d.cyw" % (os.path.splitext(os.path.basename(weights))[0], st.st_size, int(st.st_mtime)))
        if not os.path.exists(path):
            tmp = path + ".%d.tmp" % os.getpid()
            pt_import.convert_pt_to_cyw(weights, tmp)
            os.replace(tmp, path)
        return path

    def engine(self, device=None):
        if self._det is None:
            dev = device if device is not None else (self._dev if self._dev is not None else
                                                     int(os.environ.get("LOCAL_RANK", "0")))
            self._det = HipDetector(self._wpath, device=dev, **self._kw)
        return self._det

    def __call__(self, image, device=None, imgsz=640, conf=0.25, iou=0.7, **ignored):
        det = self.engine(device)
        img = np.asarray(image, dtype=np.float64)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("expected an (H,W,3) image")
        planes = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))[None]).to(det.tdev)
        netin, lb = det.letterbox_pack(planes, int(imgsz))
        pred = det.forward(netin)
        d, _, cnt = det.decode_nms(pred, lb.H, lb.W, img.shape[0], img.shape[1], float(conf), float(iou))
        n = int(cnt[0].item())
        return [Results(d[0, :n])]

    def predict_tiles(self, device_mosaic, tile_coords, pre_cfg, device=None, imgsz=640, conf=0.25, iou=0.7,
                      merge_overlap_iou_thr_soft=0.3, merge_overlap_iou_thr_hard=0.8, **ignored):
        """Batched entry of the tile queue (SURVEY §8b "additive" entry; no counterpart in ultralytics).

        device_mosaic : [ny,nx] fp32 tensor on the GPU (HipDetector.mosaic_to_device)
        tile_coords   : B rows (xmin, xmax_excl, ymin, ymax_excl) like utils.generate_tiles, all of ONE shape
        pre_cfg       : cy_preproc_cfg (DataPreprocessor.program())
        Runs TileTask.find_sources' per-tile chain (caesar_yolo/inference.py:173-275: crop, preprocessing, model call,
        process_detections) for the whole batch; returns one Results per tile (boxes in TILE pixel coordinates), or None
        where the reference would skip the tile (pipeline gave None / constant rows)."""
        det = self.engine(device)
        tc = np.asarray(tile_coords, dtype=np.int64).reshape(-1, 4)
        if len(tc) == 0:
            return []
        tw, th = tc[:, 1] - tc[:, 0], tc[:, 3] - tc[:, 2]
        if (tw != tw[0]).any() or (th != th[0]).any():
            raise ValueError("predict_tiles needs tiles of one shape per call (group ragged edge tiles separately)")
        d, cnt, status = det.detect_tiles(device_mosaic, [(int(r[0]), int(r[2])) for r in tc], int(th[0]), int(tw[0]),
                                          int(imgsz), pre_cfg, float(conf), float(iou),
                                          float(merge_overlap_iou_thr_soft), float(merge_overlap_iou_thr_hard))
        cnt, status = cnt.cpu().numpy(), status.cpu().numpy()
        return [Results(d[b, :int(cnt[b])]) if status[b] == 0 else None for b in range(len(tc))]
