#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and HBM traffic counters of
# the default bench command.  Counters in their own runs, as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$1
mkdir -p $OUT
python3 /opt/rocm/bin/rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
python3 /opt/rocm/bin/rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --repeats 1 --plain --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
python3 /opt/rocm/bin/rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 1 --repeats 1 --plain --no-cpu-baseline > $OUT/bench_write.log 2>&1
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
res = {}
for k in sorted(set(fetch) | set(write)):
    res[k] = dict(FETCH_SIZE_KB_avg=fetch.get(k, (None, 0))[0], WRITE_SIZE_KB_avg=write.get(k, (None, 0))[0],
                  dispatches=fetch.get(k, (0, 0))[1])
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, v in res.items():
    if k.startswith("tfk_sweep"): print(k, v)
PY
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
grep '"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json      # (rocprofv3 prints after the program's last line)
cut -c1-400 $OUT/bench_under_rocprof.json
