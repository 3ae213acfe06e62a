#!/usr/bin/env python
"""Benchmark of the tiled-YOLO detect path on MI355X (driver contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[2]/[3]: a seeded synthetic 16384x16384 single-channel mosaic (caesar_yolo_amd/synth.py,
"S16k"), 512x512 tiles at step 0.8 -> 1600 tiles (1521 full, 39+39 ragged, 1 corner), --preprocessing zscale(0.25) +
minmax(0,255), imgsz 512, conf 0.7, NMS IoU 0.5, merge thresholds 0.3/0.8, yolov8l nc=5 with seeded random-init weights.
One STEP = one full pass over the mosaic's tile grid: every tile of this rank's share through crop -> preprocessing ->
letterbox/pack -> YOLOv8l forward -> decode/NMS -> IoU merge (all on device, fp16 operands / fp32 accumulate), then ONE
all-gather of detection records and the cross-tile merge into the final catalog in host memory.  The mosaic is resident in
HBM before the timed region.  value = tiles processed by all ranks / wall time (max over ranks); total work is fixed as
N grows ("strong").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_DENSE_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md, "~2.5 PF dense")
FLOP_PER_TILE_512 = 105.488e9        # algorithmic conv FLOPs per 512x512 tile, yolov8l nc=5


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(mosaic_host, grid, names_w, budget_s=22.0, max_tiles=32, budget_8t_s=8.0):
    """The CPU oracle (restated reference path: numpy preprocessing + torch-CPU fp32 YOLOv8l + NMS + IoU merge),
    sequential, batch 1 like caesar_yolo/inference.py:611-622, on a bounded sample of the same tiles."""
    import numpy as np
    import torch
    from oracle import preprocessing_ref as P
    from oracle import yolov8_ref as Y
    from oracle import postproc_ref as R
    scale, names, wd = names_w
    cores = torch.get_num_threads()
    om = Y.OracleYOLO(wd, names, scale)
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    full = [i for i, t in enumerate(grid) if t[1] - t[0] == 512 and t[3] - t[2] == 512]
    sample = full[len(full) // 3:][:max_tiles]

    def run(tids, budget):
        done, t0 = 0, time.time()
        for tid in tids:
            x0, x1, y0, y1 = grid[tid]
            tile = np.array(mosaic_host[y0:y1, x0:x1], dtype=np.float32)
            tile[~np.isfinite(tile)] = 0
            img = dp(P.to_cube(tile))
            if img is not None and not P.rows_constant(img):
                det, _, _, _ = om.predict_raw(img, 512, 0.7, 0.5)
                R.process_detections(det[:, :4].numpy(), det[:, 4].numpy(), det[:, 5].numpy(), 0.7, 0.3, 0.8)
            done += 1
            if time.time() - t0 > budget:
                break
        return done, time.time() - t0
    run(sample[:1], 60.0)                                  # untimed first tile (thread pool / allocator warm-up)
    runs = []
    for nthr, budget in ((cores, budget_s * 0.5), (8, budget_8t_s + budget_s * 0.5)):      # all cores, and 8 threads
        if nthr > cores or any(r["threads"] == nthr for r in runs):
            continue
        torch.set_num_threads(nthr)
        done, dt = run(sample, budget)
        runs.append({"threads": nthr, "tiles_per_s": done / dt, "tiles": done, "seconds": dt})
    torch.set_num_threads(cores)
    best = max(runs, key=lambda r: r["tiles_per_s"])        # the better of the two is the baseline (more threads is not faster here)
    return {"value": best["tiles_per_s"], "unit": "tiles/s", "cores": best["threads"], "kind": "port",
            "sample": "%d full 512x512 tiles of the S16k grid (tids %d..), sequential batch 1, zscale+minmax + torch-CPU fp32 "
                      "yolov8l + NMS + IoU merge, %.1f s with %d torch threads" % (best["tiles"], sample[0], best["seconds"], best["threads"]),
            "runs": runs}


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the tracked rocprofv3 PMC passes (profiles/*_hbm_traffic.json,
    collected by tools/collect_profiles.sh on this same command line); None if no profile has been recorded."""
    import glob

    def family(n):
        n = n.replace(" ", "")
        if "conv3x3_wide_kernel" in n:                     # <TAIL, WN>: the 128-channel (WN=2) and the 64-channel (WN=1) variant
            return "conv3x3_wide_kernel/64" if (",1>" in n or "WN=1" in n) else "conv3x3_wide_kernel/128"
        for fam in ("conv3x3_halo2_kernel", "conv3x3_halo_kernel", "conv3x3_pp_kernel", "conv3x3_c64_kernel", "stem_mfma_kernel"):
            if fam in n:
                return fam
        if "conv1x1_direct_kernel" in n:
            return "conv1x1_direct_kernel<4" if ("<4,2" in n or "ILi4ELi2E" in n) else "conv1x1_direct_kernel<2"
        if "conv_igemm_kernel" in n:
            if "<4,2,4,3>" in n or "Li4ELi2ELi4ELi3E" in n:
                return "conv_igemm_kernel<4,2,4,3>"
            return "conv_igemm_kernel<2,2,4>" if ("<2,2,4>" in n or "Li2ELi2ELi4E" in n) else "conv_igemm_kernel<4,1,2>"
        return None
    want, best = family(kernel_label), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        for name, v in t.items():
            if want and family(name) == want:
                best = v["hbm_bytes_per_launch"]
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=16384, help="mosaic edge (default: the 16k config)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch hipEvents in the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: WORLD_SIZE=%d, --gpus=%d; using WORLD_SIZE" % (world, args.gpus))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("CY_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on fewer GPUs than ranks
    if local >= ndev and backend == "nccl":
        raise SystemExit("rank %d has no GPU (found %d); one process per GPU is required" % (rank, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    from caesar_yolo_amd import synth, utils, weights as W
    from caesar_yolo_amd import preprocessing as PP
    from caesar_yolo_amd.model import YOLO
    from caesar_yolo_amd.inference import TileEngine

    t_setup = time.time()
    model = YOLO("seeded:l:5", precision="fp16", max_batch=args.batch, max_imgsz=512, device=local) if rank == 0 else None
    if world > 1:
        dist.barrier()                       # rank 0 wrote the seeded weight file; the others reuse it
    if model is None:
        model = YOLO("seeded:l:5", precision="fp16", max_batch=args.batch, max_imgsz=512, device=local)
    det = model.engine(local)
    mosaic_host = synth.make_mosaic(args.size, seed=20260104)
    mosaic = det.mosaic_to_device(mosaic_host)          # resident before the timed region
    grid = utils.generate_tiles(0, args.size - 1, 0, args.size - 1, 512, 512, 0.8, 0.8)
    cfg = PP.DataPreprocessor([PP.ZScaleTransformer([0.25] * 3), PP.MinMaxNormalizer(0, 255)]).program()
    eng = TileEngine(det, mosaic, grid, cfg, 512, 0.7, 0.5, 0.3, 0.8, rank, world, args.batch)
    torch.cuda.synchronize()
    if rank == 0:
        log("setup %.1f s: %d tiles, %d ranks, %d tiles on rank 0" % (time.time() - t_setup, len(grid), world, eng.n_my))

    tsplit = {"local": 0.0, "gather": 0.0, "merge": 0.0}

    def step():
        t0 = time.time()
        eng.run_local()
        torch.cuda.synchronize()
        t1 = time.time()
        eng.gather()
        torch.cuda.synchronize()
        t2 = time.time()
        src = stats = None
        if rank == 0:
            src, stats = eng.merged_records()         # final catalog records in host memory (D2H synchronises)
        t3 = time.time()
        tsplit["local"] += t1 - t0; tsplit["gather"] += t2 - t1; tsplit["merge"] += t3 - t2
        return src, stats

    for _ in range(args.warmup):
        step()
    for k in tsplit:
        tsplit[k] = 0.0
    if not args.no_profile:
        det.profile(2)                   # hipEvents around every launch of every second batch (all of them cost ~3 % of the step)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        src, stats = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof = det.profile_summary() if not args.no_profile else None
    det.profile(False)

    if rank == 0:
        ntiles = len(grid)
        value = ntiles * args.steps / dt
        out = {
            "metric": "512x512 tiles/sec over 16k x 16k FITS", "value": value, "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d 1-chan FITS (S16k recipe, seed 20260104), 512x512 tiles step 0.8 "
                                   "(%d tiles), zscale+minmax, yolov8l nc=5 seeded weights, imgsz 512, conf 0.7, iou 0.5, "
                                   "merge 0.3/0.8, per-tile IoU merge + cross-tile merge" % (args.size, args.size, ntiles),
                       "tiles": ntiles, "tile_batch": args.batch, "parallelism": "tile-sharded x%d" % world,
                       "sources_in_catalog": len(src), "tiles_skipped": stats["skipped"],
                       "merge_host_ms": stats.get("merge_host_ms"), "merge_d2h_ms": stats.get("merge_d2h_ms"),
                       "per_tile_detections": stats["per_tile_detections"]},
            "conv_stack_mfma_frac_whole_job": value * FLOP_PER_TILE_512 / 1e12 / (PEAK_FP16_DENSE_TFLOPS * world),
            "step_split_ms_rank0": {k: 1000.0 * v / args.steps for k, v in tsplit.items()},
        }
        if prof:
            prof = [p for p in prof if p["launches"]]
            k = max(prof, key=lambda p: p["ms"])          # dominant kernel = largest share of GPU time in the forward
            if k["launches"] and k["ms"] > 0:
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                out["roofline"] = {"kernel": k["kernel"], "bound": "mfma", "achieved": ach, "peak": PEAK_FP16_DENSE_TFLOPS,
                                   "unit": "TFLOP/s", "frac": ach / PEAK_FP16_DENSE_TFLOPS, "traffic": pmc_traffic(k["kernel"]),
                                   "flops_per_launch": k["flops"] / k["launches"],
                                   "avg_launch_ms": k["ms"] / k["launches"], "launches": k["launches"],
                                   "timing": "hipEvents around every launch of every second batch on the launch stream, over the timed region (rank 0)"}
            tot_ms = sum(p["ms"] for p in prof)
            out["forward_kernels"] = [{"kernel": p["kernel"], "ms_total": p["ms"], "launches": p["launches"],
                                       "TFLOP/s": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["ms"] > 0 else 0.0,
                                       "share_of_forward": p["ms"] / tot_ms if tot_ms else 0.0} for p in prof]
            out["profiled_forward_share_of_step"] = tot_ms / (1000.0 * dt) if dt > 0 else None   # only every 2nd batch is timed
            out["profile_stride_batches"] = 2
        if world == 1 and not args.no_cpu_baseline:
            scale, names, wd, _ = W.read_cyw(model._wpath)
            out["cpu_baseline"] = cpu_baseline(mosaic_host, grid, (scale, names, wd))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
