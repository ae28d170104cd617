"""PCIe-inclusive rate of the drop-in seam #1 path (host arrays in, host arrays out)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from triflow_amd import Model, workloads
name, fd, pars, dt, _ = workloads.config_inputs(3)
m = Model(*workloads.model_args(name))
f = m.fields_template(**fd)
m.F(f, pars); m.J(f, pars)
t0 = time.perf_counter(); n = 5
for _ in range(n): F = m.F(f, pars)
tF = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n): J = m.J(f, pars)
tJ = (time.perf_counter() - t0) / n
N = fd['x'].size
print("model.F: %.1f ms/call (%.2f GB/s of 48 MB moved over PCIe);  model.J (csc_matrix; data array gathered on the device, 152 MB down): %.1f ms/call" % (tF*1e3, 48e6/tF/1e9, tJ*1e3))
