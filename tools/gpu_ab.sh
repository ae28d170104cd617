#!/bin/bash
# A/B of kernel variants on the GPU box: one bench.py line per TRIFLOW_HIPCC_EXTRA setting.
# usage: tools/gpu_ab.sh <tag> "<bench args>" "<extra hipcc flags>[|ENV=val ENV2=val]" ...
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
i=0
for extra in "$@"; do
    i=$((i+1))
    echo "=== variant $i: [$extra]" | tee -a $OUT/summary.txt
    flags="${extra%%|*}"; envs=""; [[ "$extra" == *"|"* ]] && envs="${extra#*|}"
    env TRIFLOW_HIPCC_EXTRA="$flags" $envs timeout -k 10 300 python3 bench.py --no-cpu-baseline --repeats 5 $ARGS > $OUT/v$i.json 2> $OUT/v$i.err || { echo "variant failed"; tail -5 $OUT/v$i.err; }
    python3 - "$OUT/v$i.json" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernels_ms_per_step", {})
    print("value %.1f steps/s  ms/step %.4f  sweep frac %.3f  step frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"] or 0, d["roofline_step"]["frac"]))
    print("  " + "  ".join("%s %.1f" % (n.replace("tfk_", ""), v * 1e3) for n, v in sorted(k.items(), key=lambda kv: -kv[1])))
except Exception as ex:
    print("no result:", ex)
PY
done
