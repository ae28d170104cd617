#!/bin/bash
# Round 3, ninth GPU call: parity suite; step-doubling trial after the one-wait change; config 5 line with the oracle in a child process.
TAG=${1:-r3i}
OUT=gpurun_out/$TAG
mkdir -p $OUT gpurun_out/refresh_r03a
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; stop_if_killed $rc
timeout -k 10 300 python3 tools/gpu_default_path_rate.py > gpurun_out/refresh_r03a/step_doubling_trial.txt 2>&1; stop_if_killed $?
cat gpurun_out/refresh_r03a/step_doubling_trial.txt
timeout -k 10 600 python3 bench.py --config 5 --repeats 7 > gpurun_out/refresh_r03a/bench_config5.json 2> gpurun_out/refresh_r03a/bench_config5.err; stop_if_killed $?
python3 - <<PY
import json
d = json.loads(open("gpurun_out/refresh_r03a/bench_config5.json").read().strip().splitlines()[-1])
print("config5", round(d["value"], 1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("fused_frac"), d["roofline_step"]["frac"], d.get("parity"), d["cpu_baseline"])
PY
