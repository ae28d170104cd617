#!/bin/bash
# solver level-plan scan on the GPU: chunk length of level 1 (m1) and of the reduced levels (m_upper)
for cfg in "32 8" "32 6" "32 5" "32 4" "32 12" "24 6" "40 6" "48 6" "28 8" "36 8"; do
  set -- $cfg
  TRIFLOW_M1=$1 TRIFLOW_M_UPPER=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('m1=$1 m_upper=$2', d['config']['solver_levels'], round(d['value'],1), 'steps/s', round(d['ms_per_step'],3))"
done
