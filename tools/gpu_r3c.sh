#!/bin/bash
# Round 3, third GPU call: parity suite; A/B of the fused level-1 back-substitution and the tail kernel; stamps.
TAG=${1:-r3c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "|TRIFLOW_L1_FUSE_BACKSUB=0" "|TRIFLOW_CR_TAIL=0" "|TRIFLOW_CR_TAIL=0 TRIFLOW_L1_FUSE_BACKSUB=0"
timeout -k 10 300 python3 tools/gpu_stamps.py > $OUT/stamps.txt 2>&1; stop_if_killed $?
cat $OUT/stamps.txt
