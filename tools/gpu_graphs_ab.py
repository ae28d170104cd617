"""HIP-graph replay of fixed steps (TRIFLOW_GRAPHS=1/0) on small grids: steps/s through the Python
scheme protocol and through the Ensemble loop (one ctypes call per step), and the final states
of both settings compared bit for bit.  Run once per setting; the second run compares."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from triflow_amd import Model, schemes, workloads
from triflow_amd.ensemble import Ensemble

tag = os.environ.get("TRIFLOW_GRAPHS", "default")
out = {}
for cfg, sch in ((1, "Theta"), (2, "Theta"), (3, "ROS2")):
    for N in (200, 2000, 20000, 200000):
        if cfg == 1 and N > 200:
            continue
        name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
        m = Model(*workloads.model_args(name))
        dev = {"Theta": schemes.Theta, "ROS2": schemes.ROS2}[sch](m)
        f, t = m.fields_template(**fd), 0.0
        for _ in range(12):
            t, f = dev(t, f, dt, pars)
        solver = f._device_backing().stepper.solver
        solver.sync()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            t, f = dev(t, f, dt, pars)
        solver.sync()
        r_py = n / (time.perf_counter() - t0)
        out["cfg%d_N%d_py" % (cfg, N)] = f.uflat.copy()
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, nstate=2)
        for _ in range(12):
            ens.step(dt)
        ens.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            ens.step(dt)
        ens.sync()
        r_ens = n / (time.perf_counter() - t0)
        out["cfg%d_N%d_ens" % (cfg, N)] = ens.state().copy()
        print("graphs=%s config %d %s N=%-7d scheme protocol %8.0f steps/s   Ensemble loop %8.0f steps/s"
              % (tag, cfg, sch, N, r_py, r_ens), flush=True)
path = os.path.join(ROOT, "gpurun_out", "graphs_ab_%s.npz" % tag)
np.savez(path, **out)
other = os.path.join(ROOT, "gpurun_out", "graphs_ab_%s.npz" % ("0" if tag == "1" else "1"))
if os.path.exists(other):
    ref = np.load(other)
    same = all(np.array_equal(ref[k], out[k]) for k in out)
    print("final states with and without graphs bit-identical:", same)
    if not same:
        for k in out:
            if not np.array_equal(ref[k], out[k]):
                print("  differs:", k, np.abs(ref[k] - out[k]).max())
