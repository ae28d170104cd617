#!/bin/bash
# Round 3: the stiff model's factorisation walk by two wavefronts with ONE exchange slot (25 KB: four workgroups per CU)
TAG=${1:-r3t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "M5_stiff or config_steps or full_size_step_config5 or respike or split or hook_input" > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "-DTF_L1_SPLIT=0"
