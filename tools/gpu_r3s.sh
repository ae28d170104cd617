#!/bin/bash
# Round 3: a step with a hook reads its input in place when the slot already satisfies the hook: test, A/B on config 5
TAG=${1:-r3s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "hook or dirichlet or steps_golden or config_steps" > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_HOOK_IN_PLACE=0"
