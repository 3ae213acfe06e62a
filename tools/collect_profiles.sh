#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root).  Writes raw rocprofv3 output under
# gpurun_out/; tools/summarize_profiles.py turns it into the tracked files under profiles/.
# usage: tools/collect_profiles.sh <tag> [s16k|c5]
set -e
TAG=${1:-r02}
CFG=${2:-s16k}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/${TAG}_ktrace $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write $R/gpurun_out/${TAG}_pmc_sq
python3 $R/bench.py --config $CFG > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_ktrace -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
echo "pmc write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -- python3 $R/bench.py --config $CFG --size 8192 --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
ls $R/gpurun_out/${TAG}_*
