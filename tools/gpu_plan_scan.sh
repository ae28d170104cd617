#!/bin/bash
# solver level-plan scan on the GPU: chunk length of level 1 (m1) and of the reduced levels (m_upper)
for cfg in "32 16" "28 16" "24 16" "20 16" "16 16" "40 16" "32 8" "24 8"; do
  set -- $cfg
  TRIFLOW_M1=$1 TRIFLOW_M_UPPER=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('m1=$1 m_upper=$2', d['config']['solver_levels'], round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'l1', round(sum(v for n,v in k.items() if n.startswith('tfk_l1')),3), 'cr', round(sum(v for n,v in k.items() if n.startswith('tfk_cr')),3))"
done
