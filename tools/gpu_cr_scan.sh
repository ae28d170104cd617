#!/bin/bash
# where to switch from chunk walks to cyclic reduction (total nodes of a reduced level)
for t in 0 1000 4000 8192 16000 40000 300000; do
  for extra in "" "--members-per-gpu 8" "--config 5"; do
    TRIFLOW_CR_MAX_NODES=$t timeout -k 10 150 python bench.py --no-cpu-baseline --steps 30 $extra 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cr_max_nodes=$t $extra', d['config']['solver_levels'], round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
