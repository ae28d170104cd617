#!/bin/bash
# FETCH_SIZE of the sweep-like kernels for two settings of an environment knob (x2: the
# counter tallies half the bytes, tools/fetch_calib.hip).  usage: gpu_pmc_ab.sh "ENV=a" "ENV=b" [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for v in "$A" "$B"; do
  OUT=gpurun_out/pmcab_$(echo $v | tr -c 'A-Za-z0-9' '_'); rm -rf $OUT; mkdir -p $OUT
  env $v rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py --steps 5 --warmup 1 --repeats 1 --plain --no-cpu-baseline "$@" > $OUT/log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("tfk_"):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("$v:", {k.replace("tfk_", ""): round(2 * sum(v) / len(v) * 1024 / 1e6, 1) for k, v in sorted(agg.items()) if "sweep" in k or "spmv" in k or "berr" in k}, "MB read per launch")
PY
done
