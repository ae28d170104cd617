#!/bin/bash
# Runs on the GPU box: kernel trace of a short bench run; prints the duration of every
# solver launch by grid size (which level it was).  usage: gpu_trace_levels.sh [tag] [bench args...]
# (TRIFLOW_* variables of the caller reach the traced run)
set -o pipefail
TAG=${1:-trace_levels}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
python3 /opt/rocm/bin/rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY | tee $OUT/levels.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if name.startswith(("tfk_cr", "tfk_bt", "tfk_top", "tfk_l1", "tfk_s_")):
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
            agg[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg, key=lambda k: (k[0], -int(k[1]) if k[1].isdigit() else 0)):
    v = sorted(agg[k]); print("%-20s grid %-9s n=%-4d median %8.2f us  min %8.2f" % (k[0], k[1], len(v), v[len(v)//2], v[0]))
PY
rm -rf $OUT/trace
