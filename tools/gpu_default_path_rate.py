"""Row f1: one trial of the step-doubling controller that Simulation(...) wraps around every
scheme (simulation.py:190-197, schemes.py:33-66): a coarse step 10*dt, ten fine steps dt and
the norm of the difference.  Fused (tf_step_doubling, one host wait) against the same trial
driven step by step from Python, and against 11 plain fixed steps; config 2 (diffusion,
N = 1e6, Theta) and config 3 (film, N = 1e6, ROS2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from triflow_amd import Model, schemes, workloads
from triflow_amd.device import null_hook

for cfg in (2, 3):
    name, fd, pars, dt, sch = workloads.config_inputs(cfg)
    model = Model(*workloads.model_args(name))
    make = {"Theta": schemes.Theta, "ROS2": schemes.ROS2}[sch]
    scheme, f, t = make(model), model.fields_template(**fd), 0.0
    for _ in range(5):
        t, f = scheme(t, f, dt, pars)
    solver = f._device_backing().stepper.solver
    solver.sync()
    n = 220 if cfg == 2 else 66
    t0 = time.perf_counter()
    for _ in range(n):
        t, f = scheme(t, f, dt, pars)
    solver.sync()
    fixed = (time.perf_counter() - t0) / n
    ntr = 12
    # fused trials
    g, err = schemes._fused_trial(scheme, t, f, dt / 10, 10, pars, null_hook, 2)
    t0 = time.perf_counter()
    for _ in range(ntr):
        g, err = schemes._fused_trial(scheme, t, g, dt / 10, 10, pars, null_hook, 2)
    fused = (time.perf_counter() - t0) / ntr
    # the same trial from Python: 11 scheme calls + the device norm
    def python_trial(fields):
        _, coarse = scheme(t, fields, 10 * (dt / 10), pars)
        tt = t
        for _ in range(10):
            tt, fields = scheme(tt, fields, dt / 10, pars)
        e = max(schemes._difference_norms(coarse, fields, 2)) / 99
        return fields, e
    g2, e2 = python_trial(f)
    c0 = solver.counters()
    t0 = time.perf_counter()
    for _ in range(ntr):
        g2, e2 = python_trial(g2)
    py = (time.perf_counter() - t0) / ntr
    c1 = solver.counters()
    print("   the %d trials driven from Python made %d factorisation(s) and %d synchronising check(s)"
          % (ntr, c1["factorisations"] - c0["factorisations"], c1["checks"] - c0["checks"]), flush=True)
    print("config %d (%s, N=%d): fixed step %.3f ms -> 11 steps %.3f ms; step-doubling trial: fused %.3f ms "
          "(%.2f x 11 steps), driven from Python %.3f ms (%.2f x)"
          % (cfg, sch, fd["x"].size, fixed * 1e3, 11 * fixed * 1e3, fused * 1e3, fused / (11 * fixed),
             py * 1e3, py / (11 * fixed)), flush=True)
