#!/bin/bash
# Round 3, sixth GPU call: parity suite; proportional Jacobian entries A/B; wide-model rates; 8 members.
TAG=${1:-r3f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "-DTF_USE_JALIAS=0"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" ""
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "-DTF_USE_JALIAS=0"
for model in wide4 six five5; do
  for gate in kernel object; do
    TRIFLOW_SPILL_GATE=$gate timeout -k 10 400 python3 tools/gpu_wide_rates.py $model 2>&1 | tail -1 | tee -a $OUT/wide.txt; stop_if_killed ${PIPESTATUS[0]}
  done
  TRIFLOW_ALLOW_SCRATCH=1 timeout -k 10 400 python3 tools/gpu_wide_rates.py $model 2>&1 | tail -1 | tee -a $OUT/wide.txt; stop_if_killed ${PIPESTATUS[0]}
done
