#!/bin/bash
# Per-kernel register / scratch / LDS usage of one HIP source, as the compiler reports it (developer tool).
# usage: tools/kernel_resources.sh caesar_yolo_amd/csrc/cy_preproc.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Name:| VGPRs:|AGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - - | sed 's/  */ /g'
