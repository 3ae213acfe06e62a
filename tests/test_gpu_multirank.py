"""Driver-visible multi-rank run of the real HIP path on ONE card: `bench.py --gpus 2` with CY_BENCH_BACKEND=gloo starts two
ranks (fresh child processes through torch.distributed.run: never an exec from this pytest process), both on GPU 0; the
collective of the tile records runs over gloo instead of RCCL, everything else -- the partition, per-rank mosaic regions,
the HIP kernels, the record layout, the cross-tile merge on rank 0 -- is the N > 1 path of caesar_yolo/inference.py:936-984's
replacement.  The catalog must be the catalog of the N = 1 run of the same build, made in the same test (same digest over the
final records, same counts)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, env):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-profile", "--no-exclusive", "--parity-steps", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_gives_the_n1_catalog():
    env = dict(os.environ)
    env["CY_BENCH_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    one = _bench(1, env)
    out = _bench(2, env)
    assert one["n_gpus"] == 1 and one["config"]["tiles"] == 1600
    assert out["n_gpus"] == 2 and out["config"]["backend"] == "gloo" and out["config"]["tiles"] == 1600
    for k in ("per_tile_detections", "sources_in_catalog", "tiles_skipped", "catalog_sha1"):
        assert out["config"][k] == one["config"][k], "%s: %r with two ranks, %r with one" % (k, out["config"][k], one["config"][k])
    assert 15000 <= one["config"]["per_tile_detections"] <= 21000 and 8000 <= one["config"]["sources_in_catalog"] <= 10500
    assert len(out["per_rank"]) == 2 and sum(r["tiles"] for r in out["per_rank"]) == 1600
    assert all(r["tiles"] > 700 and r["mosaic_mb"] < 700 for r in out["per_rank"])       # each rank holds about half of the 1074 MB mosaic
    print("two ranks on one GPU (gloo): %.0f tiles/s, %d per-tile detections, %d sources (N = 1: the same, digest %s), per rank %s" % (
        out["value"], out["config"]["per_tile_detections"], out["config"]["sources_in_catalog"], one["config"]["catalog_sha1"][:12], out["per_rank"]))


def test_record_gather_through_rccl_gives_the_n1_catalog():
    """The collective of the N > 1 path on the real library: `CY_BENCH_FORCE_DIST=1` gives the N = 1 run a process group of ONE rank over
    RCCL (backend nccl), so TileEngine.gather() issues `all_gather_into_tensor` through RCCL on the GPU, behind cy_detect_flush and the
    stream-ordered record copies, with RCCL's own stream alive beside the context's four.  Same catalog digest as the plain run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CY_BENCH_BACKEND"):
        env.pop(k, None)
    plain = _bench(1, env)
    env["CY_BENCH_FORCE_DIST"] = "1"
    rccl = _bench(1, env)
    assert plain["config"]["backend"] == "none" and rccl["config"]["backend"] == "nccl" and rccl["n_gpus"] == 1
    for k in ("per_tile_detections", "sources_in_catalog", "tiles_skipped", "catalog_sha1"):
        assert rccl["config"][k] == plain["config"][k], "%s: %r through RCCL, %r without" % (k, rccl["config"][k], plain["config"][k])
    print("one-rank RCCL group: %.0f tiles/s (plain %.0f), gather %.3f ms, digest %s" % (
        rccl["value"], plain["value"], rccl["step_split_ms_rank0"]["gather"], rccl["config"]["catalog_sha1"][:12]))
