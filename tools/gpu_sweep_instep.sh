#!/bin/bash
# in-step duration of the F+J sweep (HIP events inside bench.py) for several sweep geometries
for cfg in "8 64" "4 64" "16 64" "8 128" "8 256" "4 128" "4 256" "16 128" "2 256"; do
  set -- $cfg
  TRIFLOW_SWEEP_SEG=$1 TRIFLOW_SWEEP_BLOCK=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seg=$1 block=$2 sweep_fj', round(d['roofline']['avg_launch_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],3), 'steps/s', round(d['value'],1))"
done
