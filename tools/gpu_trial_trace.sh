#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trial_trace; rm -rf $OUT; mkdir -p $OUT
python3 /opt/rocm/bin/rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 tools/gpu_trial_trace.py > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
# the last trial: from the last tfk_diffnorm but one to the last one
idx = [i for i, e in enumerate(ev) if e[2].startswith("tfk_diffnorm")]
a, b = idx[-2], idx[-1]
prev_end = ev[a][1]
print("one trial, from the end of the previous norm kernel: %.1f us" % ((ev[b][1] - ev[a][1]) / 1e3))
busy = 0
for s, e, n in ev[a + 1:b + 1]:
    print("  gap %6.1f us  %-28s %6.1f us" % ((s - prev_end) / 1e3, n[:28], (e - s) / 1e3))
    busy += e - s
    prev_end = e
print("busy %.1f us" % (busy / 1e3))
PY
