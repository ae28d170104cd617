#!/bin/bash
# Round 3, first GPU call: parity suite, then the A/B of the stored-pivot-order block inversion
# (tf_gj_node) and of the fused tail launch (tfk_cr_tail), in-kernel stamps, per-level trace.
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; stop_if_killed $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; stop_if_killed ${PIPESTATUS[0]}
bash tools/gpu_ab.sh $TAG "--steps 50" "" "-DTF_GJ_STATIC=0|TRIFLOW_CR_TAIL=0" "-DTF_GJ_STATIC=0" "|TRIFLOW_CR_TAIL=0"
timeout -k 10 300 python3 tools/gpu_stamps.py > $OUT/stamps.txt 2>&1; stop_if_killed $?
TRIFLOW_HIPCC_EXTRA=-DTF_GJ_STATIC=0 timeout -k 10 300 python3 tools/gpu_stamps.py > $OUT/stamps_search.txt 2>&1; stop_if_killed $?
cat $OUT/stamps.txt $OUT/stamps_search.txt
timeout -k 10 400 bash tools/gpu_trace_levels.sh > $OUT/levels.txt 2>&1; cat $OUT/levels.txt
