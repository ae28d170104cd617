#!/bin/bash
# The GPU calls of round 4, one stage per call: gpurun -- bash tools/gpu_r4.sh <stage>   (stages: a b c d e f g h i j)
# Every stage writes under gpurun_out/; profiles/r04_ab_runs.txt and the other r04_* files quote them.
case "$1" in
a)
    # round 4, first call: tests on the tree as it stands, a bench line of the box, and the three
    # measurement-only items of the round-3 review (sweep context, default user path, rescue cost)
    O=gpurun_out/r4a; mkdir -p $O
    python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
    timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
    timeout -k 10 300 python3 tools/gpu_sweep_context.py > $O/sweep_context.txt 2>&1; tail -4 $O/sweep_context.txt
    timeout -k 10 200 python3 tools/gpu_simulation_rate.py --nodes 20000 --iters 2 > $O/sim_small.txt 2>&1; tail -6 $O/sim_small.txt
    timeout -k 10 600 python3 tools/gpu_simulation_rate.py --iters 3 > $O/sim_cfg3.txt 2>&1; tail -6 $O/sim_cfg3.txt
    timeout -k 10 300 python3 tools/gpu_simulation_rate.py --config 2 --iters 3 > $O/sim_cfg2.txt 2>&1; tail -4 $O/sim_cfg2.txt
    timeout -k 10 900 python3 tools/gpu_rescue_cost.py > $O/rescue.txt 2>&1; tail -30 $O/rescue.txt
    ;;
b)
    # round 4: the fused cyclic-reduction factorisation (tf_cr3_hip.h) against the round-3 kernel
    O=gpurun_out/r4b; mkdir -p $O
    python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_solve or factorisation or block or config_steps or smoke or rescue or unstable or tail" > $O/pytest_solver.log 2>&1; tail -5 $O/pytest_solver.log
    grep -q "failed" $O/pytest_solver.log && exit 1
    bash tools/gpu_ab.sh r4b_cfg3 "" "" "-DTF_CR_V4=0"
    bash tools/gpu_ab.sh r4b_cfg5 "--config 5" "" "-DTF_CR_V4=0"
    bash tools/gpu_ab.sh r4b_m8 "--members-per-gpu 8" "" "-DTF_CR_V4=0"
    timeout -k 10 300 python3 tools/gpu_stamps.py > $O/stamps.txt 2>&1; tail -25 $O/stamps.txt
    python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
    ;;
c)
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
    ;;
d)
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
    ;;
e)
    # round 4: full GPU suite; scalar solve with agent-scope hand-off; the Theta / BDF-2 probe on config 5
    O=gpurun_out/r4e; mkdir -p $O
    python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    grep -q "failed" $O/pytest.log && exit 1
    bash tools/gpu_ab.sh r4e_cfg2 "--config 2" "" "|TRIFLOW_S_FUSE=0" "|TRIFLOW_REUSE_FACTOR=0"
    bash tools/gpu_trace_levels.sh r4e_trace_cfg2 --config 2 > /dev/null; cat gpurun_out/r4e_trace_cfg2/levels.txt
    bash tools/gpu_ab.sh r4e_cfg5 "--config 5" ""
    bash tools/gpu_ab.sh r4e_cfg3 "" ""
    timeout -k 10 300 python3 tools/gpu_small_n.py > $O/small_n.txt 2>&1; tail -12 $O/small_n.txt
    ;;
f)
    # round 4: b = 8 through the round-4 factorisation; wide-model rates; stamps of the factorisation at -O3;
    # dispersive models at larger c / dx^p; the default user path
    O=gpurun_out/r4f; mkdir -p $O
    python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_solve or factorisation or block or wide or b8" > $O/pytest_solver.log 2>&1; tail -4 $O/pytest_solver.log
    grep -q "failed" $O/pytest_solver.log && exit 1
    for m in wide4 six five5; do timeout -k 10 400 python3 tools/gpu_wide_rates.py $m 2>/dev/null | tail -1 | tee -a $O/wide.txt; done
    TRIFLOW_ALLOW_SCRATCH=1 timeout -k 10 300 python3 tools/gpu_stamps.py > $O/stamps.txt 2>&1; grep -A3 "^level [2-5]" $O/stamps.txt
    RESCUE_STEPS=30 timeout -k 10 900 python3 tools/gpu_rescue_cost.py > $O/rescue.txt 2>&1; grep -c steps/s $O/rescue.txt; grep "1e+08\|1e+10\|FAILED\|rescue" $O/rescue.txt
    timeout -k 10 600 python3 tools/gpu_simulation_rate.py --iters 8 > $O/sim_cfg3.txt 2>&1; tail -5 $O/sim_cfg3.txt
    ;;
g)
    # round 4: scalar walks with four rows / eight nodes in flight
    O=gpurun_out/r4g; mkdir -p $O
    python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scalar_solve or linear_solve or config_steps or steps_golden or drift or small or tiny or constant" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    grep -q "failed" $O/pytest.log && exit 1
    bash tools/gpu_ab.sh r4g_cfg2 "--config 2" "" "|TRIFLOW_REUSE_FACTOR=0"
    bash tools/gpu_trace_levels.sh r4g_trace_cfg2 --config 2 > /dev/null; cat gpurun_out/r4g_trace_cfg2/levels.txt
    timeout -k 10 300 python3 tools/gpu_small_n.py > $O/small_n.txt 2>&1; tail -8 $O/small_n.txt
    bash tools/gpu_ab.sh r4g_cfg3 "" ""
    ;;
h)
    # stage passes with a compile-time number of stage vectors; the split walk at -O1
    O=gpurun_out/r4h; mkdir -p $O
    python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "steps_golden or fused_stage or split_factorisation or row_monitor or simulation" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    grep -q "failed" $O/pytest.log && exit 1
    bash tools/gpu_ab.sh r4h_rodaspr "--scheme RODASPR" ""
    bash tools/gpu_ab.sh r4h_ros3prl "--scheme ROS3PRL" ""
    bash tools/gpu_ab.sh r4h_cfg3 "" ""
    ;;
i)
    # the next trial queued ahead of the error read; new state + error estimate in one pass
    O=gpurun_out/r4i; mkdir -p $O
    python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "landing or scalar_solve or steps_golden or simulation or errors" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    grep -q "failed" $O/pytest.log && exit 1
    timeout -k 10 600 python3 tools/gpu_simulation_rate.py --iters 20 > $O/sim_cfg3.txt 2>&1; tail -4 $O/sim_cfg3.txt
    timeout -k 10 300 python3 tools/gpu_simulation_rate.py --config 2 --iters 20 > $O/sim_cfg2.txt 2>&1; tail -4 $O/sim_cfg2.txt
    ;;
j)
    # config 2 at N = 2e5: 19 800 steps/s in round 2, 9 800 in r4e -- which change?
    bash tools/gpu_ab.sh r4j_cfg2_2e5 "--config 2 --nodes 200000" "" "|TRIFLOW_S_FUSE=0" "|TRIFLOW_M1=16" "|TRIFLOW_M1=4"
    bash tools/gpu_trace_levels.sh r4j_trace --config 2 --nodes 200000 > /dev/null; cat gpurun_out/r4j_trace/levels.txt
    bash tools/gpu_ab.sh r4j_cfg5 "--config 5" "" "|TRIFLOW_M1=48" "|TRIFLOW_M1=64"
    ;;
*) echo "usage: $0 <a|b|c|d|e|f|g|h|i|j>"; exit 2;;
esac
