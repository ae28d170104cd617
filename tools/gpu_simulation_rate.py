"""What a user who changes only the import gets: ``Simulation(model, fields, pars, dt)`` with its
defaults -- adaptive RODASPR (embedded-error control, one host wait per trial for ``err``,
schemes.py:176-238) inside the universal step-doubling wrapper (schemes.py:33-66,
simulation.py:190-197) -- on BASELINE config 3 at N = 1e6 (or ``--config 2``), against the
fixed-step schemes of the same model.  Counts what an accepted ``dt`` costs: Rosenbrock steps
(tf_step_row calls), factorisations, synchronising checks, norm downloads."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                      # noqa: E402
from triflow_amd import Model, Simulation, schemes, workloads          # noqa: E402
from triflow_amd import _capi                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--nodes", type=int, default=0)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--dt", type=float, default=0.0)
ap.add_argument("--tol", type=float, default=1e-1)
args = ap.parse_args()

name, fd, pars, dt, fixed_scheme = workloads.config_inputs(args.config, args.nodes or None)
dt = args.dt or dt
model = Model(*workloads.model_args(name))
N = fd["x"].size

calls = dict(row=0, theta=0, norm=0, err_waits=0)
_row, _theta, _norm = _capi.DeviceSolver.step_row, _capi.DeviceSolver.step_theta, _capi.DeviceSolver.diff_norms
_rowq, _rerr = _capi.DeviceSolver.step_row_queued, _capi.DeviceSolver.read_err


def step_row(self, *a, **k):
    calls["row"] += 1
    if k.get("want_err", True) and (len(a) > 6 and a[6] is not None or k.get("b_pred") is not None):
        calls["err_waits"] += 1
    return _row(self, *a, **k)


def step_row_queued(self, *a, **k):
    calls["row"] += 1
    return _rowq(self, *a, **k)


def read_err(self, *a, **k):
    calls["err_waits"] += 1
    return _rerr(self, *a, **k)


def step_theta(self, *a, **k):
    calls["theta"] += 1
    return _theta(self, *a, **k)


def diff_norms(self, *a, **k):
    calls["norm"] += 1
    return _norm(self, *a, **k)


_capi.DeviceSolver.step_row, _capi.DeviceSolver.step_theta, _capi.DeviceSolver.diff_norms = step_row, step_theta, diff_norms
_capi.DeviceSolver.step_row_queued, _capi.DeviceSolver.read_err = step_row_queued, read_err


def solver_of(fields):
    return fields._device_backing().stepper.solver


def fixed_rate(make, n):
    scheme, f, t = make(model), model.fields_template(**fd), 0.0
    for _ in range(3):
        t, f = scheme(t, f, dt, pars)
    s = solver_of(f)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        t, f = scheme(t, f, dt, pars)
    s.sync()
    return n / (time.perf_counter() - t0)


def simulation(label, iters, **kw):
    sim = Simulation(model, dict(fd), pars, dt, **kw)
    it = iter(sim)
    t0 = time.perf_counter()
    next(it)                                   # first accepted dt: includes the start-up of the controllers
    s = solver_of(sim.fields)
    s.sync()
    first = time.perf_counter() - t0
    for k in calls:
        calls[k] = 0
    c0 = s.counters()
    t0 = time.perf_counter()
    for _ in range(iters):
        next(it)
    s.sync()
    wall = time.perf_counter() - t0
    c1 = s.counters()
    steps = calls["row"] + calls["theta"]
    print("%-46s first dt %.3f s; then %.3f s per accepted dt = %.2f accepted dt/s; per accepted dt: "
          "%.1f scheme steps (%.3f ms each), %.1f factorisations, %.1f backward-error checks, "
          "%.1f waits for the embedded error, %.1f norm downloads"
          % (label, first, wall / iters, iters / wall, steps / iters, 1e3 * wall / max(steps, 1),
             (c1["factorisations"] - c0["factorisations"]) / iters, (c1["checks"] - c0["checks"]) / iters,
             calls["err_waits"] / iters, calls["norm"] / iters), flush=True)
    return iters / wall


print("config %d: %s, N = %d, dt = %g, tol = %g" % (args.config, name, N, dt, args.tol), flush=True)
if args.config == 2:
    print("fixed-step Theta: %.0f steps/s" % fixed_rate(lambda m: schemes.Theta(m), 300), flush=True)
    simulation("Simulation(scheme=Theta) [step doubling]", args.iters, scheme=schemes.Theta, tol=args.tol)
else:
    print("fixed-step ROS2: %.0f steps/s" % fixed_rate(lambda m: schemes.ROS2(m), 100), flush=True)
    print("fixed-step RODASPR: %.0f steps/s" % fixed_rate(lambda m: schemes.RODASPR(m, time_stepping=False), 60), flush=True)
    simulation("Simulation(scheme=ROS2) [step doubling]", args.iters, scheme=schemes.ROS2, tol=args.tol)
simulation("Simulation(...) default: adaptive RODASPR in step doubling", args.iters, tol=args.tol)
simulation("Simulation(..., time_stepping=False): fixed RODASPR", 10 * args.iters, time_stepping=False)
