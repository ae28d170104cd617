#!/bin/bash
# Round 3, second GPU call: parity suite, A/B of the prefetching tail kernel and the adaptive check interval.
TAG=${1:-r3b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -rP > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; stop_if_killed $rc
grep -h "rel err vs oracle" $OUT/pytest.log
bash tools/gpu_ab.sh $TAG "--steps 50" "" "|TRIFLOW_CR_TAIL=0"
timeout -k 10 400 bash tools/gpu_trace_levels.sh > $OUT/levels.txt 2>&1; cat $OUT/levels.txt
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; stop_if_killed $?
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"], 1), d["ms_per_step"], d["roofline"]["frac"], d["roofline_step"]["frac"], d.get("parity"), d["cpu_baseline"]["value"])
PY
