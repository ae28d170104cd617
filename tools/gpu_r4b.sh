#!/bin/bash
# round 4: the fused cyclic-reduction factorisation (tf_cr3_hip.h) against the round-3 kernel
O=gpurun_out/r4b; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_solve or factorisation or block or config_steps or smoke or rescue or unstable or tail" > $O/pytest_solver.log 2>&1; tail -5 $O/pytest_solver.log
grep -q "failed" $O/pytest_solver.log && exit 1
bash tools/gpu_ab.sh r4b_cfg3 "" "" "-DTF_CR_V4=0"
bash tools/gpu_ab.sh r4b_cfg5 "--config 5" "" "-DTF_CR_V4=0"
bash tools/gpu_ab.sh r4b_m8 "--members-per-gpu 8" "" "-DTF_CR_V4=0"
timeout -k 10 300 python3 tools/gpu_stamps.py > $O/stamps.txt 2>&1; tail -25 $O/stamps.txt
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
