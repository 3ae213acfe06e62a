#!/usr/bin/env python
"""Benchmark of the tiled-YOLO detect path on MI355X (driver contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config s16k|c5]

With N > 1 and no WORLD_SIZE in the environment this process only spawns the ranks (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`, one process per GPU over RCCL; it replaces
`mpirun -np N ... run.py` of the reference, test/run_inference_parallel.sh:47-52) BEFORE touching the GPU, relays rank 0's JSON
line and exits with the children's status.  Launched under torch.distributed.run it is one of those ranks.

Workload "s16k" = BASELINE.json configs[2]/[3]: a seeded synthetic 16384x16384 single-channel mosaic (caesar_yolo_amd/synth.py),
512x512 tiles at step 0.8 -> 1600 tiles (1521 full, 39+39 ragged, 1 corner), --preprocessing zscale(0.25) + minmax(0,255),
imgsz 512, conf 0.7, NMS IoU 0.5, merge thresholds 0.3/0.8, yolov8l nc=5 with seeded random-init weights.
Workload "c5" = BASELINE.json configs[4] at one-GPU size: the S32k recipe at 16384x16384, --chan3_preproc 3-channel preprocessing,
640x640 tiles at step 0.8 -> 1024 tiles (961 full + 63 ragged; the 32k mosaic has 3969 + 127), imgsz 640 (reported under its own
metric name; the default and the headline is s16k).

One STEP = one full pass over the mosaic's tile grid: every tile of this rank's share through crop -> preprocessing ->
letterbox/pack -> YOLOv8l forward -> decode/NMS -> IoU merge (all on device, fp16 operands / fp32 accumulate), then ONE
all-gather of detection records and the cross-tile merge into the final catalog in host memory.  The rank's part of the mosaic
is resident in HBM before the timed region (FITS read + H2D + byte swap are reported separately as `ms_ingest`).
value = tiles processed by all ranks / wall time (max over ranks); total work is fixed as N grows ("strong").
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the first HIP call: see caesar_yolo_amd/__init__.py (stream budget)
sys.path.insert(0, ROOT)

PEAK_FP16_DENSE_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md, "~2.5 PF dense")
WORKLOADS = {
    # name: mosaic edge, seed, tile, step, imgsz, algorithmic conv FLOPs per full tile (yolov8l nc=5)
    "s16k": dict(size=16384, seed=20260104, tile=512, step=0.8, imgsz=512, flop=105.488e9, batch=256,
                 metric="512x512 tiles/sec over 16k x 16k FITS", pre="zscale+minmax"),
    "c5": dict(size=16384, seed=20260105, tile=640, step=0.8, imgsz=640, flop=164.825e9, batch=256,
               metric="640x640 chan3 tiles/sec over 16k x 16k FITS (config 5 at one-GPU size)", pre="chan3+minmax"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="s16k")
    ap.add_argument("--size", type=int, default=0, help="mosaic edge (default: the workload's)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--weights", default="seeded:l:5",
                    help="seeded:<scale>:<nc> = YOLOv8 (default: yolov8l nc=5, the headline), seeded11:<scale>:<nc> = YOLO11 (SURVEY 8 f3; "
                         "reported under its own metric name); random-init either way: no trained weights ship with the reference")
    ap.add_argument("--precision", default="fp16", help="context precision of the headline run: fp16 | fp16x3 | fp32")
    ap.add_argument("--parity-steps", type=int, default=-1,
                    help="timed passes of the same workload in the fp16x3 parity context after the headline run (reported as "
                         "`parity_mode`; default: 1 when the headline is fp16, else 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch hipEvents in the timed region")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra forward passes after the timed region that time the dominant kernel alone (roofline.achieved_exclusive)")
    ap.add_argument("--dry-run", action="store_true",
                    help="ranks only rendezvous, partition the grid and run the (empty) all-gather on CPU tensors: exercises the "
                         "launch path on a box without GPUs (tests/test_bench_spawn.py)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: that rank exits 3
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args, argv):
    """Parent of an N-rank run: never initialises the GPU.  Returns the exit status."""
    import socket
    import __graft_entry__ as ge
    ge.build(load=False)                                   # compile once, before N ranks race for it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("bench: spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            log(ln)
    rc = p.wait()
    if rc != 0:
        log("bench: a rank failed (launcher exit status %d)" % rc)
        return rc if 0 < rc < 256 else 1
    if line is None:
        log("bench: the ranks printed no result line")
        return 1
    if json.loads(line).get("n_gpus") != args.gpus:
        log("bench: result line reports n_gpus=%r, %d were asked for" % (json.loads(line).get("n_gpus"), args.gpus))
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(mosaic_host, grid, names_w, wl, budget_s=24.0, max_tiles=48, arch="yolov8l"):
    """The CPU oracle (restated reference path: numpy preprocessing + torch-CPU fp32 YOLOv8l + NMS + IoU merge),
    sequential, batch 1 like caesar_yolo/inference.py:611-622, on a bounded sample of the same tiles, at several thread
    counts (the best is reported; `cores` = the threads it used)."""
    import numpy as np
    import torch
    from caesar_yolo_amd import pipelines as CP
    from oracle import preprocessing_ref as P
    from oracle import yolov8_ref as Y
    from oracle import postproc_ref as R
    scale, names, wd = names_w
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 8
    if arch.startswith("yolo11"):
        from oracle import yolo11_ref as O11
        om = Y.OracleYOLO(None, names, net=O11.Net11({k: (torch.from_numpy(v[0]), torch.from_numpy(v[1])) for k, v in wd.items()}, scale, len(names)))
    else:
        om = Y.OracleYOLO(wd, names, scale)
    dp = P.build_pipeline(CP.SPECS[wl["pre"]])
    ts = wl["tile"]
    full = [i for i, t in enumerate(grid) if t[1] - t[0] == ts and t[3] - t[2] == ts]
    sample = full[len(full) // 3:][:max_tiles]

    def run(tids, budget):
        done, t0 = 0, time.time()
        for tid in tids:
            x0, x1, y0, y1 = grid[tid]
            tile = np.array(mosaic_host[y0:y1, x0:x1], dtype=np.float32)
            tile[~np.isfinite(tile)] = 0
            img = dp(P.to_cube(tile)) if np.any(tile) else None
            if img is not None and not P.rows_constant(img):
                det, _, _, _ = om.predict_raw(img, wl["imgsz"], 0.7, 0.5)
                R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), 0.7, 0.3, 0.8)
            done += 1
            if time.time() - t0 > budget:
                break
        return done, time.time() - t0
    run(sample[:1], 60.0)                                  # untimed first tile (thread pool / allocator warm-up)
    counts = [n for n in (8, 16, 32, 64) if n <= max(avail, 8)]
    runs = []
    for nthr in counts:
        torch.set_num_threads(nthr)
        done, dt = run(sample, budget_s / len(counts))
        runs.append({"threads": nthr, "tiles_per_s": done / dt, "tiles": done, "seconds": dt})
    best = max(runs, key=lambda r: r["tiles_per_s"])
    return {"value": best["tiles_per_s"], "unit": "tiles/s", "cores": best["threads"], "kind": "port",
            "host_cores_available": avail,
            "sample": ("%d full %dx%d tiles of the grid (tids %d..), sequential batch 1, %s + torch-CPU fp32 yolov8l + NMS + IoU "
                       "merge, %.1f s with %d torch threads" % (best["tiles"], ts, ts, sample[0], wl["pre"], best["seconds"], best["threads"])).replace("yolov8l", arch),
            "runs": runs}


def catalog_digest(cat):
    """sha1 over the final catalog records (x1,y1,x2,y2,score,class,edge,merged as float64, catalog order): equal digests = equal catalogs."""
    import hashlib
    import numpy as np
    return hashlib.sha1(np.ascontiguousarray(np.asarray(cat, np.float64)).tobytes()).hexdigest()


def kernel_family(n):
    """rocprofv3 kernel name (demangled `conv3x3_wide_kernel<true, 2, false, 2, false>` or mangled `_ZN2cy...ILb1ELi2E...`) or one
    of the variant labels of the library's profile summary -> the key both agree on, or None."""
    import re
    n = n.replace(" ", "")
    base = None
    for b in ("conv3x3_widep_kernel", "conv3x3_wide_kernel", "conv1x1_direct_kernel", "conv_igemm_kernel", "conv3x3_halo2_kernel", "conv3x3_halo_kernel",
              "conv3x3_pp_kernel", "conv3x3_c64_kernel", "stem_down2_kernel", "stem_down_kernel", "stem_mfma_kernel", "pool5_kernel"):
        if b in n:
            base = b
            break
    if base is None:
        return None
    tail = n.split(base, 1)[1]
    if tail.startswith("<") and ">" in tail and "=" not in tail.split(">")[0]:
        targs = [a for a in tail[1:].split(">")[0].split(",") if a]
        vals = [{"true": 1, "false": 0}.get(a, int(a) if a.lstrip("-").isdigit() else None) for a in targs]
        if None in vals:
            vals = None                             # e.g. the label "conv3x3_wide_kernel<dual> ..."
    elif tail.startswith("I"):                      # mangled template arguments: Lb1E / Li4E / DF16_ / f
        vals = [int(v) for v in re.findall(r"L[bi](\d+)E", tail.split("EEv")[0])]
    else:
        vals = None                                 # a variant label of the profile summary
    if base == "conv3x3_wide_kernel":
        if vals is None:
            return base + ("/64" if "WN=1" in tail else "/dual" if "dual" in tail else "/128")
        wn, dual, split = (vals + [0, 0, 0, 0, 0])[1], (vals + [0] * 5)[2], (vals + [0] * 5)[4]
        return base + ("/64" if wn == 1 else "/dual" if dual else "/128") + ("/x3" if split else "")
    if base == "conv1x1_direct_kernel":
        if vals is None:
            return base + ("/256" if tail.startswith("<4,2>") else "/128")
        return base + ("/256" if vals[0] == 4 else "/128") + ("/x3" if (vals + [0] * 5)[4] else "")
    if base == "conv_igemm_kernel":
        if vals is None:
            vals = [int(v) for v in re.findall(r"\d+", tail.split(">")[0])]
        v = [x for x in vals if x is not None]
        return base + "<%s>" % ",".join(str(x) for x in v[:3]) + (",3" if len(v) > 3 and v[3] == 3 else "")
    return base


def pmc_traffic(kernel_label, lane=None, pattern="*_final_hbm_traffic.json"):
    """HBM bytes per launch of a kernel family from the tracked rocprofv3 PMC passes (profiles/*_hbm_traffic.json, collected by
    tools/collect_profiles.sh on this same command line; the newest file that knows the family wins; several template
    instances of one family -- e.g. the 1x1 and the strided-3x3 form of the pixels-direct kernel -- are averaged by launches).
    lane = "main": the average over the full-batch launches only (dispatches of >= 1000 workgroups; tools/summarize_profiles.py
    splits the PMC rows by grid size), the launch set `achieved` is timed on; files without the split fall back to all launches.
    -> (bytes or None, file it came from, launch set: "main" | "all")."""
    import glob
    want, best, src, which = kernel_family(kernel_label), None, None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        for use in ((lane, None) if lane else (None,)):
            tot = n = 0.0
            for name, v in t.items():
                if want and kernel_family(name) == want:
                    e = v.get(use) if use else v
                    if e:
                        tot += e["hbm_bytes_per_launch"] * e["launches"]
                        n += e["launches"]
            if n:
                best, src, which = tot / n, os.path.relpath(f, ROOT), (use or "all")
                break
    return best, src, which


# ------------------------------------------------------------------------------------------------ one rank
def dry_run(args, wl, rank, world):
    """No GPU: rendezvous over gloo, the deterministic partition, one all-gather of empty records on CPU tensors."""
    import torch
    import torch.distributed as dist
    from caesar_yolo_amd import utils
    from caesar_yolo_amd.inference import partition_tiles
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.fail_rank:
        sys.exit(3)
    size = args.size or wl["size"]
    grid = utils.generate_tiles(0, size - 1, 0, size - 1, wl["tile"], wl["tile"], wl["step"], wl["step"])
    parts = partition_tiles(grid, wl["imgsz"], world, args.batch or wl["batch"])
    mine = torch.tensor([float(sum(len(t) for _, t in parts[rank]))])
    counts = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(counts, mine)
    if rank == 0:
        print(json.dumps({"metric": wl["metric"], "value": 0.0, "unit": "tiles/s", "dry_run": True,
                          "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": 0, "warmup": 0,
                          "tiles_per_rank": [int(c.item()) for c in counts], "tiles": len(grid)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args):
    wl = dict(WORKLOADS[args.config])
    wparts = args.weights.split(":")
    y11 = wparts[0] == "seeded11"
    arch = ("yolo11" if y11 else "yolov8") + (wparts[1] if len(wparts) > 1 else "l")
    default_model = args.weights == "seeded:l:5"
    if not default_model:
        wl["metric"] = wl["metric"] + " (%s)" % arch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench: WORLD_SIZE=%d but --gpus=%d: refusing to report a line for the wrong rank count" % (world, args.gpus))
        return 2
    if args.dry_run:
        dry_run(args, wl, rank, world)
        return 0
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ndev = torch.cuda.device_count()
    backend = os.environ.get("CY_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on fewer GPUs than ranks
    if local >= ndev and backend == "nccl":
        log("rank %d has no GPU (found %d); one process per GPU is required" % (rank, ndev))
        return 2
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    # CY_BENCH_FORCE_DIST=1 at N = 1: a process group of ONE rank, so that the record all-gather really goes through the collective
    # library (RCCL) and its stream lives beside the context's -- the N > 1 stream budget on a one-GPU box (tests/test_gpu_multirank.py)
    forced = world == 1 and os.environ.get("CY_BENCH_FORCE_DIST") == "1"
    if forced:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 or forced:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = dict(rank=0, world_size=1) if forced else {}
        if backend == "nccl" and os.environ.get("CY_BENCH_LAZY_PG") != "1":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), **kw)
        elif backend == "nccl":                # (experiment: the communicator and its stream are created by the first collective)
            dist.init_process_group("nccl", **kw)
        else:
            dist.init_process_group(backend, **kw)
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    from caesar_yolo_amd import synth, utils, weights as W
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.inference import TileEngine, MosaicSource
    from caesar_yolo_amd import pipelines as CP
    from caesar_yolo_amd.lib import letterbox as utils_letterbox

    size = args.size or wl["size"]
    # tiles per launch: 256 in the fp16 context; the fp32 / fp16x3 contexts hold 4 bytes per activation value and the largest
    # tensor of a launch must stay below the 4 GiB a buffer resource can address -> 128
    batch = args.batch or (wl["batch"] if args.precision == "fp16" else min(wl["batch"], 128))
    t_setup = time.time()
    # rank 0 writes the seeded weight file and the synthetic FITS; the others reuse both
    import tempfile
    fits_path = os.path.join(tempfile.gettempdir(), "cy_bench_%s_%d_%d.fits" % (args.config, size, os.getuid()))
    mosaic_host = None
    if rank == 0:
        YOLO(args.weights)             # resolves (= writes on first use) the seeded weight file before the other ranks look for it
        mosaic_host = synth.make_mosaic(size, seed=wl["seed"])
        utils.write_fits_image(fits_path, mosaic_host, synth.FITS_CARDS)
    if world > 1:
        dist.barrier()
    grid = utils.generate_tiles(0, size - 1, 0, size - 1, wl["tile"], wl["tile"], wl["step"], wl["step"])
    cfg = CP.device_pipeline(wl["pre"]).program()

    def make_engine(precision, nbatch):
        """context + this rank's share of the mosaic resident in HBM -> (model, detector, engine, source, ingest ms)"""
        m = YOLO(args.weights, precision=precision, max_batch=nbatch, max_imgsz=wl["imgsz"], device=local)
        d = m.engine(local)
        # ingest, timed: FITS header + memory map -> this rank's regions -> H2D -> byte swap / non-finite -> 0 on device
        torch.cuda.synchronize()
        t_in = time.time()
        data, _hdr = utils.read_fits_image(fits_path)
        sr = MosaicSource(data, big_endian=True)
        en = TileEngine(d, sr, grid, cfg, wl["imgsz"], 0.7, 0.5, 0.3, 0.8, rank, world, nbatch)
        torch.cuda.synchronize()
        return m, d, en, sr, 1000.0 * (time.time() - t_in)

    model, det, eng, src, ms_ingest = make_engine(args.precision, batch)
    # the same ingest once more: the first one also pays for the allocations (pinned staging buffers, the device buffer of the
    # band) that a process ingesting one mosaic after the other pays once
    t_in = time.time()
    src_w = MosaicSource(utils.read_fits_image(fits_path)[0], big_endian=True)
    eng_w = TileEngine(det, src_w, grid, cfg, wl["imgsz"], 0.7, 0.5, 0.3, 0.8, rank, world, batch)
    torch.cuda.synchronize()
    ms_ingest_warm = 1000.0 * (time.time() - t_in)
    del eng_w, src_w
    if rank == 0:
        log("setup %.1f s: %d tiles, %d ranks, %d tiles on rank 0, ingest %.1f ms (%.1f MB on device)" % (
            time.time() - t_setup, len(grid), world, eng.n_my, ms_ingest, src.bytes_uploaded / 1e6))

    if not default_model:
        # algorithmic FLOPs of a full tile as the runtime counts them per launch (2 x MACs of every convolution incl. depth-wise ones)
        lbx = utils_letterbox(wl["tile"], wl["tile"], wl["imgsz"])
        xprobe = torch.rand((2, lbx.H, lbx.W, 4), device="cuda").to(det.dtype)
        det.profile(True)
        det.forward(xprobe)
        torch.cuda.synchronize()
        wl["flop"] = sum(p["flops"] for p in det.profile_summary()) / 2.0
        det.profile(False)
        del xprobe
    tsplit = {"local": 0.0, "gather": 0.0, "merge": 0.0}

    def step(e=None):
        e = e or eng
        t0 = time.time()
        e.run_local()
        torch.cuda.synchronize()
        t1 = time.time()
        e.gather()
        torch.cuda.synchronize()
        t2 = time.time()
        cat = stats = None
        if rank == 0:
            cat, stats = e.merged_records()           # final catalog records in host memory (D2H synchronises)
        t3 = time.time()
        tsplit["local"] += t1 - t0; tsplit["gather"] += t2 - t1; tsplit["merge"] += t3 - t2
        return cat, stats

    for _ in range(args.warmup):
        step()
    for k in tsplit:
        tsplit[k] = 0.0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(args.steps):
        if k == args.steps - 1 and not args.no_profile:
            det.profile(1)               # hipEvents around EVERY launch of the last timed step (all batches, both lanes): ~3 % of that step
        cat, stats = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    per_rank = None
    if world > 1:
        cpu = dist.get_backend() != "nccl"
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if cpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        mine = torch.tensor([tsplit["local"], tsplit["gather"], ms_ingest, float(eng.n_my), float(src.bytes_uploaded)],
                            dtype=torch.float64, device="cpu" if cpu else "cuda")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in t.cpu()] for t in allr]
    # main lane (the full batches on the launch stream: 95 % of the tiles) and all launches incl. the small-batch lane, whose
    # kernels run on their own stream BESIDE the main lane's (their event times and the main lane's overlap)
    prof = det.profile_summary(0) if not args.no_profile else None
    prof_all = {p["kernel"]: p for p in det.profile_summary(-1) if p["launches"]} if not args.no_profile else None
    det.profile(False)
    # beside the in-situ figure: the same kernels timed with the GPU to themselves (one full batch through cy_forward on the
    # launch stream, no preprocessing / post-processing / second batch beside it), after the timed region
    prof_excl = None
    if prof and world == 1 and not args.no_exclusive:
        lbx = utils_letterbox(wl["tile"], wl["tile"], wl["imgsz"])
        x = torch.rand((batch, lbx.H, lbx.W, 4), device="cuda").to(det.dtype)
        det.forward(x)
        torch.cuda.synchronize()
        det.profile(True)
        for _ in range(2):
            det.forward(x)
        torch.cuda.synchronize()
        prof_excl = {p["kernel"]: p for p in det.profile_summary() if p["launches"]}
        det.profile(False)
        del x

    tsplit_main = dict(tsplit)                               # (the parity passes below go through the same step())
    # ---- the same workload in the parity context (fp16x3: the mode that meets the north star's parity bar), same timing protocol
    parity = None
    psteps = args.parity_steps if args.parity_steps >= 0 else (1 if args.precision == "fp16" else 0)
    if psteps > 0:
        del eng
        pb = min(batch, 128)                                   # 4 bytes per activation value: 128 tiles keep every tensor below 4 GiB
        pmodel, pdet, peng, psrc, _ = make_engine("fp16x3", pb)
        step(peng)                                             # warm-up pass
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tp0 = time.time()
        for _ in range(psteps):
            pcat, pstats = step(peng)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        pdt = time.time() - tp0
        if world > 1:
            cpu = dist.get_backend() != "nccl"
            tmax = torch.tensor([pdt], dtype=torch.float64, device="cpu" if cpu else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            pdt = float(tmax.item())
        if rank == 0:
            n2, n3 = pdet.weight_passes()
            parity = {"dtype": "fp16x3", "precision": "activations as fp16 high + low halves (22 significand bits), fp32 accumulate.  Layers whose filter is "
                                                      "exactly fp16 values x a per-channel factor (a checkpoint stored in fp16 with its BatchNorm folded in "
                                                      "fp32, as ultralytics loads it) run TWO MFMA passes (x_lo*w, x_hi*w), any other filter three "
                                                      "(x_lo*w_hi, x_hi*w_lo, x_hi*w_hi).  The context tests/test_gpu_configs.py holds to the north star's "
                                                      "parity bar: kept-anchor sets identical to the fp32 oracle's, boxes within 1e-4 of the image size "
                                                      "(<= 0.06 px), scores 2e-5",
                      "layers_two_pass": n2, "layers_three_pass": n3,
                      "value": len(grid) * psteps / pdt, "unit": "tiles/s", "ms_per_step": 1000.0 * pdt / psteps, "steps": psteps, "warmup": 1,
                      "tile_batch": pb, "sources_in_catalog": len(pcat), "per_tile_detections": pstats["per_tile_detections"],
                      "vs_fp16_headline": None}
        del peng, psrc
        pdet.close()

    if rank == 0:
        ntiles = len(grid)
        value = ntiles * args.steps / dt
        lbf = utils_letterbox(wl["tile"], wl["tile"], wl["imgsz"])
        flops_pass = 0.0
        for t in grid:
            lbt = utils_letterbox(t[3] - t[2], t[1] - t[0], wl["imgsz"])
            flops_pass += wl["flop"] * (lbt.H * lbt.W) / float(lbf.H * lbf.W)
        if parity:
            parity["vs_fp16_headline"] = parity["value"] / value if args.precision == "fp16" else None
            npass = 2.0 if parity["layers_three_pass"] == 0 else 3.0
            parity["conv_stack_algorithmic_frac_whole_job"] = flops_pass * parity["value"] / ntiles / 1e12 / (PEAK_FP16_DENSE_TFLOPS * world)
            parity["conv_stack_mfma_frac_whole_job"] = npass * parity["conv_stack_algorithmic_frac_whole_job"]
            parity["conv_stack_mfma_frac_note"] = "MFMA work of the context = %d x the algorithmic conv FLOPs (%d fp16 passes per product); `conv_stack_algorithmic_frac_whole_job` counts each product once" % (npass, npass)
        n_gpus = dist.get_world_size() if world > 1 else 1        # the ranks the communicator actually has
        ms_step = 1000.0 * dt / args.steps
        ms_in = max(r[2] for r in per_rank) if per_rank else ms_ingest
        out = {
            "metric": wl["metric"], "value": value, "unit": "tiles/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"fp16": "f16", "fp32": "f32"}.get(args.precision, args.precision), "data": "synthetic",
            # what `value` is and is not: the fp16 context is the opt-in throughput mode (`--precision fp16`); the product default
            # (YOLO(), scripts/run.py) is the fp16x3 parity context, whose rate on the same workload is `parity_value` below
            "value_mode": ({"fp16": "fp16 context = opt-in THROUGHPUT mode (fp16 operands, fp32 accumulate): NOT parity-grade -- 1.6-2.5 % of the kept anchors and "
                                    "2-4 % of the catalog sources differ from the fp32 oracle's (tests/test_gpu_configs.py asserts >= 95 % agreement); "
                                    "the parity-grade rate of the same workload is `parity_value`",
                            "fp16x3": "fp16x3 context = the product default and the parity context (kept-anchor sets identical to the fp32 oracle's, boxes "
                                      "within 1e-4 of the image size, scores 2e-5)",
                            "fp32": "fp32 context = exact fp32 FMA chains (parity-grade)"}[args.precision]),
            "parity_value": (parity["value"] if parity else (value if args.precision in ("fp16x3", "fp32") else None)),
            "parity_unit": "tiles/s", "parity_dtype": ("fp16x3" if parity else (args.precision if args.precision in ("fp16x3", "fp32") else None)),
            "parity_vs_value": ((parity["value"] / value) if parity else (1.0 if args.precision in ("fp16x3", "fp32") else None)),
            "parity_tolerance": "kept-box index set after NMS identical to the fp32 oracle's (ties within 2e-5 of a cut counted and printed); boxes within 1e-4 "
                                "of the image size = 0.05 px at 512 / 0.064 px at 640 (measured max 3.5e-2 px; NOT 1e-4 px, which is 2.5 fp32 ulps of a "
                                "600-px coordinate); scores within 2e-5",
            "config": {"workload": "synthetic %dx%d 1-chan FITS (S%s recipe, seed %d), %dx%d tiles step %.1f (%d tiles), %s, "
                                   "%s nc=5 seeded weights, imgsz %d, conf 0.7, iou 0.5, merge 0.3/0.8, per-tile IoU merge + "
                                   "cross-tile merge" % (size, size, "32k" if args.config == "c5" else "16k", wl["seed"], wl["tile"],
                                                         wl["tile"], wl["step"], ntiles, wl["pre"], arch, wl["imgsz"]),
                       "model": arch, "conv_flops_per_full_tile": wl["flop"],
                       "tiles": ntiles, "tile_batch": batch, "parallelism": "tile-sharded x%d" % world,
                       "backend": (dist.get_backend() if dist.is_initialized() else "none"),
                       "side_streams": int(os.environ.get("CY_SIDE_STREAMS", "2")),
                       "sources_in_catalog": len(cat), "catalog_sha1": catalog_digest(cat), "tiles_skipped": stats["skipped"],
                       "merge_host_ms": stats.get("merge_host_ms"), "merge_d2h_ms": stats.get("merge_d2h_ms"),
                       "per_tile_detections": stats["per_tile_detections"],
                       "degenerate_boxes_dropped": stats.get("degenerate_boxes", 0),
                       "candidate_overflow_tiles": stats.get("cand_overflow_tiles", 0)},
            # algorithmic conv FLOPs of the pass: per tile, the full-tile figure x (letterboxed pixels / full letterboxed pixels) -- the
            # network is fully convolutional, so the 79 ragged tiles of the 16k grid count with their own (smaller) maps
            "conv_stack_mfma_frac_whole_job": flops_pass * args.steps / dt / 1e12 / (PEAK_FP16_DENSE_TFLOPS * world),
            "conv_flops_per_pass": flops_pass,
            "step_split_ms_rank0": {k: 1000.0 * v / args.steps for k, v in tsplit_main.items()},
            # host-inclusive view (never `value`): FITS memory map -> H2D of this rank's regions -> on-device byte swap
            "ms_ingest": ms_in, "ingest_mb_rank0": src.bytes_uploaded / 1e6,
            "tiles_per_s_incl_ingest": ntiles / ((ms_step + ms_in) * 1e-3),
            "ms_ingest_warm_rank0": ms_ingest_warm,     # second ingest of the same file in this process (buffers already allocated)
            "tiles_per_s_incl_ingest_warm": ntiles / ((ms_step + ms_ingest_warm) * 1e-3) if world == 1 else None,
        }
        if parity:
            out["parity_mode"] = parity
        if per_rank:
            out["per_rank"] = [{"rank": i, "tiles": int(r[3]), "local_ms": 1000.0 * r[0] / args.steps,
                                "gather_ms": 1000.0 * r[1] / args.steps, "ingest_ms": r[2], "mosaic_mb": r[4] / 1e6}
                               for i, r in enumerate(per_rank)]
        if prof:
            prof = [p for p in prof if p["launches"]]
            k = max(prof, key=lambda p: p["ms"])          # dominant kernel = largest share of GPU time in the forward
            pset = "c5" if args.config == "c5" else ("final" if default_model else "y11")      # profile set of this command line (tools/collect_profiles.sh)
            tpat = "*_%s_hbm_traffic.json" % pset
            if k["launches"] and k["ms"] > 0:
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                traffic, tsrc, tset = pmc_traffic(k["kernel"], "main", tpat)
                out["roofline"] = {"kernel": k["kernel"], "bound": "mfma", "achieved": ach, "peak": PEAK_FP16_DENSE_TFLOPS,
                                   "unit": "TFLOP/s", "frac": ach / PEAK_FP16_DENSE_TFLOPS, "traffic": traffic,
                                   "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command line, "
                                                      "2 x FETCH + WRITE; not measured in this run; average over %s)" % (
                                                          tsrc, "the full-batch launches (dispatches of >= 1000 workgroups): the launch set `achieved` is timed on"
                                                          if tset == "main" else "ALL launches of the family incl. the small-batch lane")) if tsrc else None,
                                   "traffic_launch_set": tset,
                                   "achieved_all_launches": ((prof_all[k["kernel"]]["flops"] / (prof_all[k["kernel"]]["ms"] * 1e-3) / 1e12)
                                                             if prof_all and k["kernel"] in prof_all and prof_all[k["kernel"]]["ms"] > 0 else None),
                                   "launches_all": (prof_all[k["kernel"]]["launches"] if prof_all and k["kernel"] in prof_all else None),
                                   "flops_per_launch_all": ((prof_all[k["kernel"]]["flops"] / prof_all[k["kernel"]]["launches"])
                                                            if prof_all and k["kernel"] in prof_all and prof_all[k["kernel"]]["launches"] else None),
                                   "traffic_note": "`traffic_launch_set` = main: bytes and `achieved` refer to the same launches (compare with flops_per_launch); "
                                                   "= all: PMC average over every launch of the family (compare with flops_per_launch_all)",
                                   "achieved_exclusive": ((prof_excl[k["kernel"]]["flops"] / (prof_excl[k["kernel"]]["ms"] * 1e-3) / 1e12)
                                                          if prof_excl and k["kernel"] in prof_excl and prof_excl[k["kernel"]]["ms"] > 0 else None),
                                   "achieved_exclusive_note": "same kernel, all its launches of one full batch through the forward pass alone on the GPU "
                                                              "(after the timed region; `achieved` is measured inside the pipelined pass, beside the side "
                                                              "streams and the small-batch lane)",
                                   "flops_per_launch": k["flops"] / k["launches"],
                                   "avg_launch_ms": k["ms"] / k["launches"], "launches": k["launches"],
                                   "timing": "hipEvents around every launch of the forward pass during the last timed step (rank 0), each on the stream it runs on.  `achieved`: the launches of the full batches (main lane, 95 % of the tiles); `achieved_all_launches`: those plus the launches of the small-batch lane, which run on another stream beside them (the two sets of event times overlap, so their sum exceeds the step)"
                                             + ("; batches of 64..239 tiles run their forward as two concurrent half-batches on two streams, so the "
                                                "launches timed here overlap each other and `achieved` understates the kernel's exclusive rate by up to 2x"
                                                if 64 <= batch < 240 else "")}
            # the largest HBM-bound kernel of the forward pass: bytes per launch from the tracked PMC passes x its launches / its event time
            hb = None
            for q in prof:
                fam = kernel_family(q["kernel"]) or ""
                if fam in ("conv3x3_c64_kernel", "conv1x1_direct_kernel/128", "stem_down2_kernel", "stem_down_kernel") and q["ms"] > 0:
                    byt, bsrc, bset = pmc_traffic(q["kernel"], "main", tpat)
                    if byt and (hb is None or q["ms"] > hb[0]["ms"]):
                        hb = (q, byt, bsrc, bset)
            if hb:
                q, byt, bsrc, bset = hb
                if bset != "main" and prof_all and q["kernel"] in prof_all:
                    q = prof_all[q["kernel"]]           # bytes averaged over ALL launches (a persistent kernel's launches cannot be split by grid size): time over all launches too
                tbs = byt * q["launches"] / (q["ms"] * 1e-3) / 1e12
                out["roofline_hbm"] = {"kernel": q["kernel"], "bound": "hbm", "achieved": tbs, "peak": 8.0, "unit": "TB/s", "frac": tbs / 8.0,
                                       "traffic": byt, "launches": q["launches"], "avg_launch_ms": q["ms"] / q["launches"],
                                       "traffic_source": "%s (2 x FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes, averaged over %s; launch "
                                                         "times: hipEvents of this run over the same launch set)" % (
                                                             bsrc, "the full-batch launches (same launch set as the times)" if bset == "main" else
                                                             "ALL launches incl. the small-batch lane (a persistent kernel's launches cannot be told apart by grid size)"),
                                       "traffic_launch_set": bset,
                                       "TFLOP/s": q["flops"] / (q["ms"] * 1e-3) / 1e12}
            # preprocessing / decode / NMS / merge against the HBM roof: tracked figures of the same command line (PMC bytes over
            # kernel-trace durations, tools/summarize_profiles.py); these kernels run on the side streams and are not event-timed here
            import glob as _glob
            sk = sorted(_glob.glob(os.path.join(ROOT, "profiles", "*_%s_side_kernels.json" % pset)))
            if sk:
                try:
                    out["side_kernels_hbm"] = {"source": os.path.relpath(sk[-1], ROOT) + " (not measured in this run)", "peak_TB_per_s": 8.0,
                                               "kernels": json.load(open(sk[-1]))}
                except Exception:
                    pass
            tot_ms = sum(p["ms"] for p in prof)
            out["forward_kernels"] = [{"kernel": p["kernel"], "ms_total": p["ms"], "launches": p["launches"],
                                       "TFLOP/s": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["ms"] > 0 else 0.0,
                                       "share_of_forward": p["ms"] / tot_ms if tot_ms else 0.0} for p in prof]
            out["profiled_forward_share_of_step"] = tot_ms / ms_step if ms_step > 0 else None   # > 1 is possible: the small-batch lane runs beside the main lane
            out["profiled_steps"] = 1
        if world == 1 and not args.no_cpu_baseline:
            if y11:
                _g, wd = W.seeded11_folded(wparts[1], int(wparts[2]), int(wparts[3]) if len(wparts) > 3 else 11)
                scale, names = wparts[1], model.names
            else:
                scale, names, wd, _ = W.read_cyw(model._wpath)
            out["cpu_baseline"] = cpu_baseline(mosaic_host, grid, (scale, names, wd), wl, arch=arch)
        print(json.dumps(out), flush=True)
        try:
            os.remove(fits_path)
        except OSError:
            pass
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        log("--gpus must be >= 1")
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
