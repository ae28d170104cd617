#!/bin/bash
# round 4: full GPU suite; scalar solve with agent-scope hand-off; the Theta / BDF-2 probe on config 5
O=gpurun_out/r4e; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
grep -q "failed" $O/pytest.log && exit 1
bash tools/gpu_ab.sh r4e_cfg2 "--config 2" "" "|TRIFLOW_S_FUSE=0" "|TRIFLOW_REUSE_FACTOR=0"
bash tools/gpu_trace_levels.sh r4e_trace_cfg2 --config 2 > /dev/null; cat gpurun_out/r4e_trace_cfg2/levels.txt
bash tools/gpu_ab.sh r4e_cfg5 "--config 5" ""
bash tools/gpu_ab.sh r4e_cfg3 "" ""
timeout -k 10 300 python3 tools/gpu_small_n.py > $O/small_n.txt 2>&1; tail -12 $O/small_n.txt
