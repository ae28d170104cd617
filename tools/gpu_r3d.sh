#!/bin/bash
# Round 3, fourth GPU call: parity suite; level-1 chunk length scan with the fused back-substitution; cr_factor block.
TAG=${1:-r3d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -8 $OUT/pytest.log; stop_if_killed $rc
bash tools/gpu_ab.sh $TAG "--steps 50" "" "|TRIFLOW_M1=24" "|TRIFLOW_M1=28" "|TRIFLOW_M1=20" "|TRIFLOW_M1=36" "|TRIFLOW_CR_FACTOR_BLOCK=512"
