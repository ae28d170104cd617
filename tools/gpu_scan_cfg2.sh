for cfg in "32 6" "32 12" "32 24" "32 48" "64 24"; do
  set -- $cfg
  TRIFLOW_M1=$1 TRIFLOW_M_UPPER=$2 timeout -k 10 120 python bench.py --config 2 --no-cpu-baseline --steps 100 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('m1=$1 m_upper=$2', d['config']['solver_levels'], round(d['value'],1), 'steps/s', round(d['ms_per_step'],4))"
done
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
