#!/bin/bash
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
