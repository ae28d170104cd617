"""the F+J sweep's roofline fraction slid 0.684 -> 0.669 -> 0.644 over three driver
runs although the kernel's ISA is unchanged since round 2 (profiles/r04_sweep_context.txt: the
disassemblies differ in one kernel-argument offset).  On ONE box, the duration of tfk_sweep_fj
(config 3, N = 1e6; HIP events on that kernel only)
  (b) back to back, nothing else on the GPU (inputs and outputs of the last sweep still in the
      256 MB cache);
  (c) each sweep behind a pass that streams 1 GiB through the memory system (cold caches);
  (d) inside ROS2 steps (what bench.py reports): behind the last back-substitution of the step before;
  (e) behind each of the other kernels of a step in isolation is not separable -- instead: inside
      Theta-form steps of the same model (other predecessor, same sweep body + theta epilogue is a
      different kernel: skipped)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402
import torch                                                    # noqa: E402
from triflow_amd import Model, workloads                        # noqa: E402
from triflow_amd.ensemble import Ensemble                       # noqa: E402

name, fd, pars, dt, scheme = workloads.config_inputs(3)
model = Model(*workloads.model_args(name))
fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
ens = Ensemble(model, fd["x"], fields, pars, True, scheme="ROS2", nstate=3)
s = ens.solver
K = "tfk_sweep_fj"


def timed(fn, n):
    s.timing(kernels=[K])
    s.timing_reset()
    fn(n)
    s.sync()
    ms, cnt = s.timing_report()[K]
    s.timing(False)
    return 1e3 * ms / cnt, cnt


def back_to_back(n):
    for _ in range(n):
        s.eval(0, with_j=True)


flush = torch.empty(1 << 27, dtype=torch.float64, device="cuda")      # 1 GiB


def cold(n):
    for _ in range(n):
        flush.add_(1.0)                      # reads and writes 1 GiB on torch's stream
        torch.cuda.synchronize()
        s.eval(0, with_j=True)
        s.sync()


def in_step(n):
    for _ in range(n):
        ens.step(dt)


for _ in range(10):
    ens.step(dt)
ens.sync()
out = []
for rep in range(3):
    b, nb = timed(back_to_back, 200)
    c, nc = timed(cold, 40)
    ens.restart()
    d, nd = timed(in_step, 300)
    ens.restart()
    out.append((b, c, d))
    print("pass %d: tfk_sweep_fj  (b) back to back %.2f us (%d)   (c) cold caches %.2f us (%d)   "
          "(d) inside ROS2 steps %.2f us (%d)   -> fractions of 8 TB/s: %.3f / %.3f / %.3f"
          % (rep, b, nb, c, nc, d, nd, 200e6 / (b * 1e-6) / 8e12,
             200e6 / (c * 1e-6) / 8e12, 200e6 / (d * 1e-6) / 8e12), flush=True)
print("device:", torch.cuda.get_device_properties(0).name, flush=True)
