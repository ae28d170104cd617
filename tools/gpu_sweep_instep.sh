for cfg in "8 64" "8 256" "16 256" "8 512" "16 128" "32 256"; do
  set -- $cfg
  TRIFLOW_SWEEP_SEG=$1 TRIFLOW_SWEEP_BLOCK=$2 timeout -k 10 150 python bench.py --no-cpu-baseline --steps 50 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('seg=$1 block=$2', round(d['value'],1), 'steps/s', 'fj', k['tfk_sweep_fj'], 'fstage', k['tfk_sweep_f_stage'], 'spmv', k['tfk_spmv'], 'berr', k['tfk_berr'], 'roof', round(d['roofline']['frac'],3))"
done
