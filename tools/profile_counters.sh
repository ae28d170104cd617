#!/bin/bash
# Runs on the GPU box (through gpurun): SQ counters of the solver kernels, one rocprofv3
# --pmc pass per counter group (the pool refuses --pmc together with tracing), then a
# per-kernel summary.  usage: tools/profile_counters.sh <tag> [bench.py arguments]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/ctr_$TAG
mkdir -p $OUT
ARGS="--steps 4 --warmup 1 --repeats 25 --no-cpu-baseline --plain $*"
pass() {   # name, counters...
    local name=$1; shift
    python3 /opt/rocm/bin/rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 \
        || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
    echo "pass $name done" >> $OUT/progress.txt
}
pass waves SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    if not k.startswith("tfk_"):
        continue
    row = {c: sum(v) / len(v) for c, v in d.items()}
    row["dispatches"] = max(len(v) for v in d.values())
    wc = row.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
            if c in row:
                row[c + "/WAVE_CYCLES"] = round(row[c] / wc, 4)
    if row.get("SQ_WAVES"):
        row["VALU_insts_per_wave"] = round(row.get("SQ_INSTS_VALU", 0) / row["SQ_WAVES"], 1)
    res[k] = row
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, {c: v[c] for c in v if "/" in c or c in ("SQ_WAVES", "VALU_insts_per_wave", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")})
PY
