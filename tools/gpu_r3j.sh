#!/bin/bash
# Round 3, tenth GPU call: parity suite (dense path for tiny grids, rescue with two factorisations); step-doubling trial; bench.
TAG=${1:-r3j}
OUT=gpurun_out/$TAG
mkdir -p $OUT gpurun_out/refresh_r03a
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
timeout -k 10 300 python3 tools/gpu_default_path_rate.py > gpurun_out/refresh_r03a/step_doubling_trial.txt 2>&1; stop_if_killed $?
cat gpurun_out/refresh_r03a/step_doubling_trial.txt
bash tools/gpu_ab.sh $TAG "--steps 50" ""
