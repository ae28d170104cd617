#!/bin/bash
# Round 3, seventh GPU call: parity suite; proportional entries applied at decode time; cr_factor at 5 waves.
TAG=${1:-r3g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "-DTF_USE_JALIAS=0" "-DTF_CR_FACTOR_WAVES=5|TRIFLOW_ALLOW_SCRATCH=1" ""
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "-DTF_USE_JALIAS=0"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "-DTF_USE_JALIAS=0"
