#!/bin/bash
# round 4: factorisation with the shared separator row and 2 wavefronts per chunk; staged reductions in tfk_s_fwd
O=gpurun_out/r4d; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_solve or factorisation or block or config_steps or scalar_solve or unstable or rescue or ensemble or shard" > $O/pytest_solver.log 2>&1; tail -4 $O/pytest_solver.log
grep -q "failed" $O/pytest_solver.log && exit 1
bash tools/gpu_trace_levels.sh r4d_trace_cfg3 > /dev/null; cat gpurun_out/r4d_trace_cfg3/levels.txt
bash tools/gpu_ab.sh r4d_cfg3 "" "" "|TRIFLOW_CR_FACTOR_BLOCK=256"
bash tools/gpu_ab.sh r4d_cfg5 "--config 5" "" "|TRIFLOW_CR_FACTOR_BLOCK=256"
bash tools/gpu_ab.sh r4d_m8 "--members-per-gpu 8" "" "|TRIFLOW_CR_FACTOR_BLOCK=256"
bash tools/gpu_ab.sh r4d_cfg2 "--config 2" "" "|TRIFLOW_M1=16" "|TRIFLOW_REUSE_FACTOR=0" "|TRIFLOW_REUSE_FACTOR=0 TRIFLOW_M1=16"
bash tools/gpu_ab.sh r4d_rodaspr "--scheme RODASPR" ""
