#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/ quotes for a round, in two calls.
#   tools/gpu_round_refresh.sh <round-tag> 1     tests, smoke, the default bench line, kernel stats + HBM counters, levels, stamps
#   tools/gpu_round_refresh.sh <round-tag> 2     SQ counters, the other BASELINE workloads, step-doubling trial
# (results under gpurun_out/refresh_<tag>/ and prof_/ctr_<tag>); then, back in the build container:
#   python3 tools/collect_round_profiles.py <round-tag> <prefix, e.g. r03>
TAG=$1; PART=${2:-1}
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
stop_if_killed() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit (rc $rc): stopping"; exit $rc; fi; }
if [ "$PART" = "1" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; tail -2 $OUT/pytest.log; stop_if_killed $rc
  timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; stop_if_killed $?
  timeout -k 10 500 bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; stop_if_killed $?
  timeout -k 10 300 bash tools/gpu_trace_levels.sh > $OUT/levels.txt 2>&1; stop_if_killed $?
  timeout -k 10 200 python3 tools/gpu_stamps.py > $OUT/stamps.txt 2>&1; stop_if_killed $?
  python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"], 1), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3), round(d["roofline_step"]["frac"], 3), d.get("parity"))
PY
  cat $OUT/levels.txt $OUT/stamps.txt
else
  timeout -k 10 700 bash tools/profile_counters.sh $TAG > $OUT/counters.log 2>&1; stop_if_killed $?
  timeout -k 10 400 python3 bench.py --config 5 --repeats 7 > $OUT/bench_config5.json 2> $OUT/bench_config5.err; stop_if_killed $?
  timeout -k 10 300 python3 bench.py --config 2 --repeats 7 > $OUT/bench_config2.json 2> $OUT/bench_config2.err; stop_if_killed $?
  timeout -k 10 300 python3 bench.py --members-per-gpu 8 --steps 20 --repeats 7 --cpu-workers 16 > $OUT/bench_members8.json 2>/dev/null; stop_if_killed $?
  timeout -k 10 300 python3 bench.py --scheme RODASPR --steps 20 --repeats 7 --no-cpu-baseline > $OUT/bench_rodaspr.json 2>/dev/null; stop_if_killed $?
  timeout -k 10 300 python3 tools/gpu_default_path_rate.py > $OUT/step_doubling_trial.txt 2>&1; stop_if_killed $?
  cat $OUT/step_doubling_trial.txt
  python3 - <<PY
import json
for n in ("bench_config5", "bench_config2", "bench_members8", "bench_rodaspr"):
    try:
        d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3),
              d["roofline"].get("fused_frac"), round(d["roofline_step"]["frac"], 3), d.get("parity"), d.get("factorising_every_step"))
    except Exception as ex:
        print(n, "no result:", ex)
PY
fi
