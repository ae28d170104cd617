"""Timeline of step-doubling trials of config 2 driven from Python (run under rocprofv3 --kernel-trace):
tools/gpu_trial_trace.sh prints where the time between the kernels goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triflow_amd import Model, schemes, workloads

name, fd, pars, dt, sch = workloads.config_inputs(2)
model = Model(*workloads.model_args(name))
scheme, f, t = schemes.Theta(model), model.fields_template(**fd), 0.0
for _ in range(5):
    t, f = scheme(t, f, dt, pars)
solver = f._device_backing().stepper.solver


def python_trial(fields):
    _, coarse = scheme(t, fields, 10 * (dt / 10), pars)
    tt = t
    for _ in range(10):
        tt, fields = scheme(tt, fields, dt / 10, pars)
    e = max(schemes._difference_norms(coarse, fields, 2)) / 99
    return fields, e


g = f
for _ in range(3):
    g, e = python_trial(g)
solver.sync()
t0 = time.perf_counter()
for _ in range(8):
    g, e = python_trial(g)
print("trial %.3f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
