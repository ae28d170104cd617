#!/bin/bash
# Round 3: BDF-2 with the history read in a state slot (three rotating slots): tests, A/B on config 5
TAG=${1:-r3w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -30; exit 1; }
timeout -k 10 300 python3 bench.py --config 5 --steps 20 --repeats 5 --no-cpu-baseline > $OUT/cfg5.json 2> $OUT/cfg5.err
python3 - <<PY
import json
d = json.loads(open("$OUT/cfg5.json").read().strip().splitlines()[-1])
k = d["kernels_ms_per_step"]
print("config 5: %.1f steps/s %.4f ms" % (d["value"], d["ms_per_step"]), d["roofline"]["frac"], d["roofline"].get("fused_frac"), d.get("parity"))
print("  " + "  ".join("%s %.1f" % (n.replace("tfk_", ""), v * 1e3) for n, v in sorted(k.items(), key=lambda kv: -kv[1])))
PY
