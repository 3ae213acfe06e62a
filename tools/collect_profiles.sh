#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root).  Writes raw rocprofv3 output under
# gpurun_out/<tag>_<name>_*; tools/summarize_profiles.py <tag> <name> turns it into the tracked files under profiles/.
# usage: tools/collect_profiles.sh <tag> [final|c5|y11]      final = default `python bench.py` (S16k, yolov8l), c5 = --config c5,
#                                                             y11 = --weights seeded11:l:5 (YOLO11l on the S16k workload)
set -e
TAG=${1:-r04}
NAME=${2:-final}
case $NAME in
  final) ARGS="" ;;
  c5)    ARGS="--config c5" ;;
  y11)   ARGS="--weights seeded11:l:5" ;;
  *) echo "unknown profile set $NAME"; exit 2 ;;
esac
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_${NAME}
rm -rf ${O}_ktrace ${O}_pmc_fetch ${O}_pmc_write ${O}_pmc_sq
python3 $R/bench.py $ARGS > ${O}_bench.json 2> ${O}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_ktrace -- python3 $R/bench.py $ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --parity-steps 0 > ${O}_bench_under_rocprof.json 2>/dev/null
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_pmc_fetch -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-profile --parity-steps 0 > /dev/null 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_pmc_write -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-profile --parity-steps 0 > /dev/null 2>&1
echo "pmc write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d ${O}_pmc_sq -- python3 $R/bench.py $ARGS --size 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --parity-steps 0 > /dev/null 2>&1
ls -d ${O}_*
