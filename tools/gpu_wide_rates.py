"""Rates of the wide models (row 'wide-model path' of VERDICT r2): steps/s of ROS2 at N nodes for a
model whose level-1 walks spill registers, under the per-kernel -O1 gate (default), the round-2 gate
(whole code object at -O1: TRIFLOW_SPILL_GATE=object) and no gate (TRIFLOW_ALLOW_SCRATCH=1).
usage: python tools/gpu_wide_rates.py <model> [N]     (one model and one gate setting per process)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import corpus                                   # noqa: E402  (inputs only)
from triflow_amd import Model                              # noqa: E402
from triflow_amd.ensemble import Ensemble                   # noqa: E402

EXTRA = {
    # 5 variables, 5-point stencils (b = 10: chunk walks on the reduced levels)
    "five5": (["-dxxxxA + k*dxxA + B*dxA", "k*dxxB - A*dxxxC", "k*dxxC + dxD*A", "k*dxxD - dxxxxD + B",
               "k*dxxG - dxxxxG + A*dxG"], list("ABCDG"), "k", None),
}
name = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6
args = EXTRA[name] if name in EXTRA else corpus.model_args(name)
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    model = Model(*args)
    rng = np.random.default_rng(0)
    x = np.linspace(0, N * 5e-3, N, endpoint=False)
    dep = list(model._dep_vars)
    fields = {v: (1.0 + 0.3 * np.cos(2 * np.pi * (j + 1) * x / x[-1]) + 0.01 * rng.standard_normal(N))[None, :]
              for j, v in enumerate(dep)}
    pars = dict(k=0.3, periodic=True)
    ens = Ensemble(model, x, fields, pars, True, scheme="ROS2", nstate=2)
warned = [str(w.message)[:60] for w in caught if "registers" in str(w.message)]
dt = 1e-5
for _ in range(3):
    ens.step(dt)
ens.sync()
steps = 30
t0 = time.perf_counter()
for _ in range(steps):
    ens.step(dt)
ens.sync()
el = time.perf_counter() - t0
s = ens.solver
s.timing(True); s.timing_reset()
for _ in range(5):
    ens.step(dt)
ens.sync()
rep = s.timing_report()
top = sorted(rep.items(), key=lambda kv: -kv[1][0])[:6]
print("%-6s N=%d gate=%s scratch=%s: %.1f steps/s (%.3f ms/step) levels %s finite=%s warned=%d | %s"
      % (name, N, os.environ.get("TRIFLOW_SPILL_GATE", "kernel"), os.environ.get("TRIFLOW_ALLOW_SCRATCH", "0"),
         steps / el, el / steps * 1e3, s.describe()["chunks"], bool(np.isfinite(ens.state()).all()), len(warned),
         "  ".join("%s %.0f" % (k.replace("tfk_", ""), v[0] / 5 * 1e3) for k, v in top)))
