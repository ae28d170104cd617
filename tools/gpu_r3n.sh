#!/bin/bash
# Round 3: config 5 with the fused back-substitution (y block of 80 KB per workgroup); two-rank rehearsal test.
TAG=${1:-r3n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -k "two_ranks or process_group or parity" > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_L1_LDS_MAX=81920" "|TRIFLOW_L1_LDS_MAX=81920 TRIFLOW_M1=30"
tail -3 gpurun_out/ab_${TAG}_cfg5/v2.err
