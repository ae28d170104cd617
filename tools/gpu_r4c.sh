#!/bin/bash
# round 4: new tests; per-level times of the two cyclic-reduction factorisations; the two-launch scalar solve
O=gpurun_out/r4c; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scalar_solve or landing or hooked_state or config_steps or doubling or constant or two_resident or theta or bdf2 or drift or small" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
grep -q "failed" $O/pytest_new.log && exit 1
bash tools/gpu_trace_levels.sh r4c_trace_v4 > /dev/null; cat gpurun_out/r4c_trace_v4/levels.txt
TRIFLOW_HIPCC_EXTRA=-DTF_CR_V4=0 bash tools/gpu_trace_levels.sh r4c_trace_v3 > /dev/null; cat gpurun_out/r4c_trace_v3/levels.txt
bash tools/gpu_ab.sh r4c_cfg2 "--config 2" "" "|TRIFLOW_S_FUSE=0" "|TRIFLOW_M1=16" "|TRIFLOW_M1=16 TRIFLOW_S_FUSE=0" "|TRIFLOW_REUSE_FACTOR=0" "|TRIFLOW_REUSE_FACTOR=0 TRIFLOW_S_FUSE=0"
bash tools/gpu_trace_levels.sh r4c_trace_cfg2 --config 2 > /dev/null; cat gpurun_out/r4c_trace_cfg2/levels.txt
timeout -k 10 600 python3 tools/gpu_simulation_rate.py --iters 6 > $O/sim_cfg3.txt 2>&1; tail -4 $O/sim_cfg3.txt
timeout -k 10 300 python3 tools/gpu_simulation_rate.py --config 2 --iters 6 > $O/sim_cfg2.txt 2>&1; tail -4 $O/sim_cfg2.txt
bash tools/gpu_ab.sh r4c_cfg5 "--config 5" ""
