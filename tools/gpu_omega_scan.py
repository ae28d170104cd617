"""Unrefined componentwise backward error of one solve, by level-1 chunk length."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from triflow_amd import Model, workloads
from tests import parity_cases as pc

for cfg in (3, 5, 2):
    for N in (200, 3000, 20000, 100000):
        name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
        m = Model(*workloads.model_args(name))
        out = []
        for m1 in (4, 8, 16, 32):
            s = pc.bound_solver(m, fd, pars, refine=0, m1=m1)
            s.eval(0, with_j=True)
            s.factor(dt * (0.5 if cfg == 3 else 1.0))
            rhs = np.random.default_rng(1).standard_normal(N * m._nvar)
            s.solve(rhs)
            s2 = pc.bound_solver(m, fd, pars, m1=m1)       # auto refine: report omega it measures
            s2.eval(0, with_j=True); s2.factor(dt * (0.5 if cfg == 3 else 1.0)); s2.solve(rhs)
            om, refined = s2.backward_error()
            out.append("m1=%d %s omega %.1e%s" % (m1, s.describe()["chunks"], om, " (refined)" if refined else ""))
        print("config %d N=%d: " % (cfg, N) + " | ".join(out), flush=True)
