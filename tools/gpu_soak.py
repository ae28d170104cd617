"""Long run of a BASELINE configuration on the GPU: state extrema every block of steps, until the
solver reports a problem or the step budget is spent.  usage: gpu_soak.py [config] [steps] [block]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triflow_amd import Model, workloads
from triflow_amd.ensemble import Ensemble

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
total = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
block = int(sys.argv[3]) if len(sys.argv) > 3 else 500
name, x, fields, pars, dt, scheme = bench.build_problem(cfg, None, bench.member_table(1, None))
model = Model(*workloads.model_args(name))
ens = Ensemble(model, x, fields, pars, bool(pars["periodic"]), scheme=scheme, nstate=2)
done = 0
t0 = time.time()
while done < total:
    try:
        for _ in range(block):
            ens.step(dt)
        ens.check()
    except RuntimeError as ex:
        print("after %d..%d steps: %s" % (done, done + block, ex))
        break
    done += block
    st = ens.state()
    print("steps %6d  t=%.3f  " % (done, done * dt) +
          "  ".join("var%d [%.4g, %.4g]" % (v, st[v].min(), st[v].max()) for v in range(st.shape[0])) +
          "  monitor %.1e  %.1fs" % (ens.solver.monitor_error(), time.time() - t0), flush=True)
