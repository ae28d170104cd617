"""Steps/s through the scheme protocol for small grids (the sizes of the reference's
own examples), device path against the CPU oracle on the same box."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from triflow_amd import Model, schemes, workloads
from oracle import numpy_path as ora

def rate(scheme, fields, pars, dt, n, sync=None):
    t = 0.0
    for _ in range(5):
        t, fields = scheme(t, fields, dt, pars)
    if sync: sync(fields)
    t0 = time.perf_counter()
    for _ in range(n):
        t, fields = scheme(t, fields, dt, pars)
    if sync: sync(fields)
    return n / (time.perf_counter() - t0)

sync = lambda f: f._device_backing().stepper.solver.sync()
for cfg, sch in ((2, "Theta"), (3, "ROS2")):
    for N in (200, 2000, 20000, 200000):
        name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
        m = Model(*workloads.model_args(name))
        mo = Model(*workloads.model_args(name), compiler=ora.numpy_compiler)
        dev = {"Theta": schemes.Theta, "ROS2": schemes.ROS2}[sch](m)
        cpu = {"Theta": ora.Theta, "ROS2": ora.ROS2}[sch](mo)
        r_dev = rate(dev, m.fields_template(**fd), pars, dt, 300, sync)
        r_cpu = rate(cpu, mo.fields_template(**fd), pars, dt, max(3, min(300, int(3e5 / N))))
        print("config %d %s N=%-7d device %8.1f steps/s   cpu oracle %8.1f steps/s   x%.1f" % (cfg, sch, N, r_dev, r_cpu, r_dev / r_cpu), flush=True)
