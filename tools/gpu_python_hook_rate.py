"""Steps/s with the README-style PYTHON hook (two node assignments) at N = 1e6 (config 2
model, clamped, Theta): applied on the device through point writes, against the declarative
DirichletHook and against no hook."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from triflow_amd import Model, schemes
from triflow_amd.device import DirichletHook

N = 10 ** 6
m = Model("k * dxxU", "U", "k")
x = np.linspace(0, 1, N)
fd = dict(x=x, U=np.cos(2 * np.pi * 5 * x))
pars = dict(k=1e-3, periodic=False)

def py_hook(t, fields, pars):
    fields.U[0] = 1
    fields.U[-1] = 0
    return fields, pars

def py_hook_host(t, fields, pars):          # slices: the state goes through the host
    fields.U[0:1] = 1
    fields.U[-1:] = 0
    return fields, pars

sch = schemes.Theta(m)
for name, kw in (("python hook", dict(hook=py_hook)), ("python hook via host", dict(hook=py_hook_host)), ("DirichletHook", dict(hook=DirichletHook(U={0: 1.0, -1: 0.0}))), ("no hook", {})):
    f, t = m.fields_template(**fd), 0.0
    for _ in range(5):
        t, f = sch(t, f, 1e-2, pars, **kw)
    float(np.asarray(f["U"])[1])          # completes the queue (and, for resident states, downloads once)
    t0 = time.perf_counter()
    nrun = 20 if "host" in name else 200
    for _ in range(nrun):
        t, f = sch(t, f, 1e-2, pars, **kw)
    b = f._device_backing()
    if b is not None:
        b.stepper.solver.sync()
    print("%-22s %8.1f steps/s   U[:2] = %s" % (name, nrun / (time.perf_counter() - t0), np.asarray(f["U"])[:2]))
