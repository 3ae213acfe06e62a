# stream budget A/B at N = 1 (developer tool, round 4): plain; with a one-rank process group (RCCL: its stream is a fifth beside the
# context's four) at the HIP runtime's default of four hardware queues and at eight (the package default since round 4); then the
# emulated N = 8 ranks with the RCCL group alive, both queue counts.  usage: bash tools/ab_streams.sh
run() {
  env $1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive --parity-steps 0 > gpurun_out/ab_s.json 2> gpurun_out/ab_s.err
  python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open("gpurun_out/ab_s.json") if l.startswith("{")][-1])
    print(sys.argv[1], "->", round(d["value"],1), d["config"]["backend"], d["config"]["side_streams"], d["config"]["catalog_sha1"][:10], d["step_split_ms_rank0"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("gpurun_out/ab_s.err").read()[-1500:])
PY
}
for v in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=4 CY_BENCH_FORCE_DIST=1" "GPU_MAX_HW_QUEUES=8 CY_BENCH_FORCE_DIST=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=4 CY_BENCH_FORCE_DIST=1 CY_SIDE_STREAMS=3" "GPU_MAX_HW_QUEUES=4 CY_BENCH_FORCE_DIST=1 CY_SIDE_STREAMS=1"; do run "$v"; done
for q in 4 8; do echo "== emulated N = 8 ranks, one-rank RCCL group alive, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q EMU_PG=1 python tools/emulate_ranks.py 8 256 2>/dev/null | tail -9; done
echo "== emulated N = 8 ranks, no process group"; python tools/emulate_ranks.py 8 256 2>/dev/null | tail -3
