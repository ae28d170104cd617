#!/bin/bash
# Round 3: state update inside the back-substitution of the last solve: tests, A/B on config 3, 8 members, config 5
TAG=${1:-r3r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?; tail -6 $OUT/pytest.log; stop_if_killed $rc
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
bash tools/gpu_ab.sh ${TAG}_cfg3 "--steps 20" "" "|TRIFLOW_FUSE_UPDATE=0"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "|TRIFLOW_FUSE_UPDATE=0"
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_FUSE_UPDATE=0"
