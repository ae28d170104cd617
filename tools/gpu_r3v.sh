#!/bin/bash
# Round 3: the fused theta / BDF-2 sweeps without the separate store of F (variant 2 = with it)
TAG=${1:-r3v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_FUSED_STORE_F=1"
bash tools/gpu_ab.sh ${TAG}_cfg2 "--steps 50 --config 2" "" "|TRIFLOW_FUSED_STORE_F=1"
