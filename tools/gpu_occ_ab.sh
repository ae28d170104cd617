#!/bin/bash
# does a second wavefront per SIMD help the level-1 solve walk (248 registers without the deep prefetch)?
for cfg in "32 -O3" "16 -O3" "32 -O3 -DTF_PREFETCH_DEEP(s)=0" "16 -O3 -DTF_PREFETCH_DEEP(s)=0"; do
  set -- $cfg
  m1=$1; shift
  for extra in "" "--members-per-gpu 8"; do
  TRIFLOW_M1=$m1 TRIFLOW_HIPCC_OPT="$*" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 $extra 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('m1=$m1 $* $extra', round(d['value'],1), {n: v for n, v in k.items() if n in ('tfk_l1_solve','tfk_l1_factor_rhs','tfk_l1_backsub')})"
  done
done
