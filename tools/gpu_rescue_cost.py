"""What a dispersion-dominated user pays: KdV (`-6 U dxU - dxxxU`) and
Kuramoto-Sivashinsky at N = 1e6, Theta and ROS2, over a range of c / dx^3 -- where block elimination
without pivoting across separators needs iterative refinement or the rescue on longer chunks
(DESIGN.md section 4.5).  Prints steps/s next to the solver's counters and the last measured
backward error, and the diffusion model of config 2 on the same box for scale."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402
from oracle import corpus                                       # noqa: E402
from triflow_amd import Model, schemes, workloads               # noqa: E402

N = int(os.environ.get("RESCUE_N", "1000000"))
STEPS = int(os.environ.get("RESCUE_STEPS", "60"))


def run(label, model, fd, pars, dt, make, steps=STEPS):
    scheme, f, t = make(model), model.fields_template(**fd), 0.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            for _ in range(4):
                t, f = scheme(t, f, dt, pars)
            s = f._device_backing().stepper.solver
            s.sync()
            c0 = s.counters()
            t0 = time.perf_counter()
            for _ in range(steps):
                t, f = scheme(t, f, dt, pars)
            s.sync()
            wall = time.perf_counter() - t0
            c1 = s.counters()
            om, refined = s.backward_error()
            u = np.asarray(f[model._dep_vars[0]])
            print("%-44s %8.0f steps/s   per step: %.2f factorisations, %.2f checks; replans so far %d; "
                  "last backward error %.1e%s; plan %s; finite %s%s"
                  % (label, steps / wall, (c1["factorisations"] - c0["factorisations"]) / steps,
                     (c1["checks"] - c0["checks"]) / steps, c1["replans"], om, " (refining)" if refined else "",
                     s.describe()["chunks"], bool(np.isfinite(u).all()),
                     "; warned: rescue" if any("rescue" in str(x.message) for x in w) else ""), flush=True)
        except RuntimeError as ex:
            print("%-44s FAILED: %s" % (label, str(ex)[:160]), flush=True)


name, fd, pars, dt, _ = workloads.config_inputs(2, N)
m2 = Model(*workloads.model_args(name))
run("config 2 diffusion Theta (for scale)", m2, fd, pars, dt, lambda m: schemes.Theta(m), 300)

for name in ("kdv", "kuramoto"):
    model = Model(*corpus.model_args(name))
    for length in (1e4, 1e3, 1e2):
        x = np.linspace(0, length, N, endpoint=False)
        dx = length / N
        U = 0.5 / np.cosh(0.5 * (x - 0.3 * length) / (length / 200)) ** 2 + 0.05 * np.cos(2 * np.pi * 3 * x / length)
        fdk = dict(x=x, U=U)
        parsk = dict(periodic=True)
        order = 3 if name == "kdv" else 4
        for ratio in (1e0, 1e2, 1e4, 1e6, 1e8, 1e10):
            dtk = ratio * dx ** order
            if dtk > 1.0:
                continue
            run("%s L=%g dx=%.0e dt=%.1e (c/dx^%d = %.0e) Theta" % (name, length, dx, dtk, order, ratio),
                model, fdk, parsk, dtk, lambda m: schemes.Theta(m))
        run("%s L=%g dx=%.0e dt=%.1e ROS2" % (name, length, dx, 1e2 * dx ** order),
            model, fdk, parsk, 1e2 * dx ** order, lambda m: schemes.ROS2(m))
