#!/bin/bash
# Round 3: level-1 factorisation walks by two wavefronts (band / right-hand sides): tests, then A/B
# against the one-wavefront form on config 3, 8 members per GPU and config 5.
TAG=${1:-r3o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?; tail -6 $OUT/pytest.log; stop_if_killed $rc
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
bash tools/gpu_ab.sh ${TAG}_cfg3 "--steps 20" "" "-DTF_L1_SPLIT=0"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "-DTF_L1_SPLIT=0"
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "-DTF_L1_SPLIT=0"
