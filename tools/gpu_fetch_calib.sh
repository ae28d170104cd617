#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/fetch_calib.hip), separate passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- tools/_bin/fetch_calib > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- tools/_bin/fetch_calib > $OUT/w.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("calib"):
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        v = v[4:]                                        # skip the first launches
        print("%-10s %-20s mean %9.1f KB = %6.1f MB per launch (%d launches)" % (ctr, k, sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
