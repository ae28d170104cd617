"""Host-side cost of issuing one step (no wait in between) against the step's time on the GPU:
Ensemble.step of config 2 (5 launches per step) and config 3 (18)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triflow_amd import Model, workloads
from triflow_amd.ensemble import Ensemble

for cfg in (2, 3):
    name, x, fields, pars, dt, scheme = bench.build_problem(cfg, None, bench.member_table(1, None))
    model = Model(*workloads.model_args(name))
    ens = Ensemble(model, x, fields, pars, bool(pars["periodic"]), scheme=scheme, nstate=3)
    for _ in range(50):
        ens.step(dt)
    ens.sync()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        ens.step(dt)
    t1 = time.perf_counter()
    ens.sync()
    t2 = time.perf_counter()
    print("config %d: %.1f us of host time per step issued, %.1f us per step until the GPU has finished"
          % (cfg, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    ens.close()
