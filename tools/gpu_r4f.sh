#!/bin/bash
# round 4: b = 8 through the round-4 factorisation; wide-model rates; stamps of the factorisation at -O3;
# dispersive models at larger c / dx^p; the default user path
O=gpurun_out/r4f; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_solve or factorisation or block or wide or b8" > $O/pytest_solver.log 2>&1; tail -4 $O/pytest_solver.log
grep -q "failed" $O/pytest_solver.log && exit 1
for m in wide4 six five5; do timeout -k 10 400 python3 tools/gpu_wide_rates.py $m 2>/dev/null | tail -1 | tee -a $O/wide.txt; done
TRIFLOW_ALLOW_SCRATCH=1 timeout -k 10 300 python3 tools/gpu_stamps.py > $O/stamps.txt 2>&1; grep -A3 "^level [2-5]" $O/stamps.txt
RESCUE_STEPS=30 timeout -k 10 900 python3 tools/gpu_rescue_cost.py > $O/rescue.txt 2>&1; grep -c steps/s $O/rescue.txt; grep "1e+08\|1e+10\|FAILED\|rescue" $O/rescue.txt
timeout -k 10 600 python3 tools/gpu_simulation_rate.py --iters 8 > $O/sim_cfg3.txt 2>&1; tail -5 $O/sim_cfg3.txt
