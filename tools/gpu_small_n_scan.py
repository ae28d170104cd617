"""Level-1 chunk length against grid size (device steps/s through the scheme protocol)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from triflow_amd import Model, schemes, workloads

def rate(scheme, fields, pars, dt, n):
    t = 0.0
    for _ in range(5):
        t, fields = scheme(t, fields, dt, pars)
    s = fields._device_backing().stepper.solver
    s.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        t, fields = scheme(t, fields, dt, pars)
    s.sync()
    return n / (time.perf_counter() - t0), s.describe()["chunks"]

for cfg, sch in ((3, "ROS2"), (2, "Theta"), (5, "BDF2")):
    for N in (200, 2000, 20000, 100000, 400000):
        out = []
        for m1 in (4, 8, 16, 32):
            os.environ["TRIFLOW_M1"] = str(m1)
            name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
            m = Model(*workloads.model_args(name))
            dev = {"Theta": schemes.Theta, "ROS2": schemes.ROS2, "BDF2": schemes.BDF2}[sch](m)
            r, chunks = rate(dev, m.fields_template(**fd), pars, dt, 200)
            out.append("m1=%d: %.0f %s" % (m1, r, chunks))
        print("config %d %s N=%-7d " % (cfg, sch, N) + " | ".join(out), flush=True)
