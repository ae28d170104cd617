#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/ quotes for a round, in one call.
#   tools/gpu_round_refresh.sh <round-tag>      (results under gpurun_out/refresh_<tag>/ and prof_/ctr_<tag>)
# then, back in the build container:  python3 tools/collect_round_profiles.py <round-tag> <prefix, e.g. r02>
TAG=$1
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
bash tools/profile_counters.sh $TAG > $OUT/counters.log 2>&1
bash tools/gpu_trace_levels.sh > $OUT/levels.txt 2>&1
python3 bench.py --config 5 --no-cpu-baseline --repeats 7 > $OUT/bench_config5.json 2>/dev/null
python3 bench.py --config 2 --no-cpu-baseline --repeats 7 > $OUT/bench_config2.json 2>/dev/null
python3 bench.py --members-per-gpu 8 --steps 20 --repeats 7 --cpu-workers 16 > $OUT/bench_members8.json 2>/dev/null
python3 bench.py --scheme RODASPR --steps 20 --repeats 7 --no-cpu-baseline > $OUT/bench_rodaspr.json 2>/dev/null
python3 tools/gpu_default_path_rate.py > $OUT/step_doubling_trial.txt 2>&1
python3 - <<PY
import json
for n in ("bench", "bench_config5", "bench_config2", "bench_members8", "bench_rodaspr"):
    d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(d["value"], 1), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3),
          d["roofline"].get("fused_frac"), round(d["roofline_step"]["frac"], 3))
PY
