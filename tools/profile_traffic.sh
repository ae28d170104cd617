#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes as the
# pool requires) and kernel-trace stats of another bench workload.
#   usage: tools/profile_traffic.sh <tag> "<bench args, e.g. --config 5>"
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic_$1; ARGS=$2
mkdir -p $OUT
python3 /opt/rocm/bin/rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
python3 /opt/rocm/bin/rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS --steps 5 --warmup 1 --repeats 1 --plain --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
python3 /opt/rocm/bin/rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS --steps 5 --warmup 1 --repeats 1 --plain --no-cpu-baseline > $OUT/bench_write.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
def table(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
fetch = table("/pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = table("/pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
dur = {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(out + "/kernel_stats.csv"))}
res = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("tfk_"):
        continue
    # (FETCH_SIZE counts half the bytes of this package's 8 B / lane loads: profiles/r02_fetch_size_calibration.txt)
    rd, wr = 2 * fetch.get(k, (0, 0))[0] * 1024, write.get(k, (0, 0))[0] * 1024
    us = dur.get(k, (None, 0))[0]
    res[k] = dict(read_MB=round(rd / 1e6, 1), written_MB=round(wr / 1e6, 1), avg_us=us,
                  TB_per_s=round((rd + wr) / us / 1e6, 2) if us else None, dispatches=fetch.get(k, (0, 0))[1])
json.dump(dict(args="$ARGS", kernels=res), open(out + "/traffic.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["avg_us"] or 0)):
    print("%-28s %8.1f us  read %8.1f MB  written %8.1f MB  %5s TB/s" % (k, v["avg_us"] or 0, v["read_MB"], v["written_MB"], v["TB_per_s"]))
PY
