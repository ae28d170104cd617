#!/bin/bash
# Round 3, eleventh GPU call: parity suite; scalar wavefront index in tfk_cr_factor A/B; stamps; levels.
TAG=${1:-r3k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "-DTF_CR_SCALAR_W=0" ""
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "-DTF_CR_SCALAR_W=0"
timeout -k 10 300 python3 tools/gpu_stamps.py > $OUT/stamps.txt 2>&1; stop_if_killed $?
cat $OUT/stamps.txt
timeout -k 10 300 bash tools/gpu_trace_levels.sh > $OUT/levels.txt 2>&1; stop_if_killed $?
cat $OUT/levels.txt
