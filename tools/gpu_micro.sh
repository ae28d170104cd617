#!/bin/bash
# Runs on the GPU box: builds and runs the microbenchmarks under tools/micro/.
OUT=gpurun_out/micro; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_issue.hip -o /tmp/fp64_issue 2>/dev/null && timeout -k 10 120 /tmp/fp64_issue | tee $OUT/fp64_issue.txt
hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip -o /tmp/hbm_ceiling 2>/dev/null && timeout -k 10 200 /tmp/hbm_ceiling | tee $OUT/hbm_ceiling.txt
