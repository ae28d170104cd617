#!/bin/bash
# Round 3, fifth GPU call: parity suite; twisted l1_solve A/B; config 2 graphs; 8 members twisted; config 5.
TAG=${1:-r3e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -8 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "|TRIFLOW_L1_SOLVE_TWIST=0"
bash tools/gpu_ab.sh ${TAG}_rodaspr "--steps 20 --scheme RODASPR" "" "|TRIFLOW_L1_SOLVE_TWIST=0"
bash tools/gpu_ab.sh ${TAG}_cfg2 "--steps 200 --config 2" "" "|TRIFLOW_GRAPHS=1"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "|TRIFLOW_L1_TWIST=1"
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" ""
