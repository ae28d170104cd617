#!/bin/bash
# Round 3, eighth GPU call: parity suite; walk-assembled separator rows A/B; pinned row requests A/B.
TAG=${1:-r3h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -rP > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
grep -h "walk-assembled" $OUT/pytest.log
bash tools/gpu_ab.sh $TAG "--steps 50" "" "|TRIFLOW_L1_FUSE_ASM=0" "-DTF_PIN=1" "-DTF_PIN=1 -DTF_USE_JALIAS=0"
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_L1_FUSE_ASM=0" "-DTF_PIN=1"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "|TRIFLOW_L1_FUSE_ASM=0" "-DTF_PIN=1"
bash tools/gpu_ab.sh ${TAG}_cfg2 "--steps 200 --config 2" "" "|TRIFLOW_L1_FUSE_ASM=0"
