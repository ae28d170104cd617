#!/bin/bash
# A/B of the walk prefetch depth (compile-time): default (deep only without spike columns), always deep, never deep
for opt in "-O3" "-O3 -DTF_PREFETCH_DEEP(s)=1" "-O3 -DTF_PREFETCH_DEEP(s)=0"; do
  TRIFLOW_HIPCC_OPT="$opt" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$opt', round(d['value'],1), 'steps/s', {n: v for n, v in k.items() if n.startswith('tfk_l1')})"
done
