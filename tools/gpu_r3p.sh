#!/bin/bash
# Round 3: in-kernel phase timing of the level-1 walks and begin / end of their workgroups (diagnostic build)
OUT=gpurun_out/r3p; mkdir -p $OUT
timeout -k 10 400 python3 tools/gpu_wgtrace.py > $OUT/wgtrace.txt 2>&1; rc=$?
cat $OUT/wgtrace.txt
exit $rc
