"""Long runs of config 3 (ROS2 and RODASPR, N = 1e6) and config 2 with the fused launches of round 4 and with the
launches they replace (TRIFLOW_L1CR_FUSE / TRIFLOW_S_FUSE = 0): the states must be the same bits after every block."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triflow_amd import Model, workloads
from triflow_amd.ensemble import Ensemble

blocks, block = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 500
for cfg, scheme_name, var in ((3, "ROS2", "TRIFLOW_L1CR_FUSE"), (3, "RODASPR", "TRIFLOW_L1CR_FUSE"), (2, "Theta", "TRIFLOW_S_FUSE")):
    name, x, fields, pars, dt, _ = bench.build_problem(cfg, None, bench.member_table(1, None))
    model = Model(*workloads.model_args(name))
    runs = []
    for fuse in ("1", "0"):
        os.environ[var] = fuse
        runs.append(Ensemble(model, x, fields, pars, bool(pars["periodic"]), scheme=scheme_name, nstate=2))
        del os.environ[var]
    for b in range(blocks):
        for e in runs:
            for _ in range(block):
                e.step(dt)
            e.check()
        a, c = runs[0].state(), runs[1].state()
        same = np.array_equal(a, c)
        print("config %d %-8s after %5d steps: %s  (max |diff| %.1e, monitor %.1e / %.1e)"
              % (cfg, scheme_name, (b + 1) * block, "same bits" if same else "DIFFERENT", np.abs(a - c).max(),
                 runs[0].solver.monitor_error(), runs[1].solver.monitor_error()), flush=True)
        if not same:
            sys.exit(1)
    for e in runs:
        e.close()
print("ok")
