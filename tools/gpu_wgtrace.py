"""When do the workgroups of the level-1 kernels start and end?  (diagnostic build -DTF_STAMPS:
every workgroup records s_memrealtime at its first and last instruction.)"""
import os, sys
os.environ["TRIFLOW_HIPCC_EXTRA"] = (os.environ.get("TRIFLOW_HIPCC_EXTRA", "") + " -DTF_STAMPS").strip()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from triflow_amd import Model, workloads
from triflow_amd.ensemble import Ensemble

members = int(sys.argv[1]) if len(sys.argv) > 1 else 1
table = bench.member_table(members, None)
name, x, fields, pars, dt, scheme = bench.build_problem(3, None, table)
model = Model(*workloads.model_args(name))
ens = Ensemble(model, x, fields, pars, True, scheme=scheme, nstate=2)
s = ens.solver
REG = 8 + 3 * 32
s.debug_stamps(REG)
for _ in range(5):
    ens.step(dt)
ens.sync()
st = s.debug_stamps(REG).astype(np.int64).reshape(-1)
for k, kname in enumerate(("tfk_l1_factor_rhs", "tfk_l1_fwd2_backsub", "tfk_l1_solve")):
    t = st[512 + 2048 * k: 512 + 2048 * (k + 1)].reshape(-1, 2)
    t = t[t[:, 0] > 0]
    if not len(t):
        continue
    t0 = t[:, 0].min()
    b, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us
    d = e - b
    print("%-22s %4d workgroups: start %.1f / %.1f / %.1f us (median / 90 %% / last), duration %.1f / %.1f / %.1f us (min / median / max), last end %.1f us"
          % (kname, len(t), np.median(b), np.percentile(b, 90), b.max(), d.min(), np.median(d), d.max(), e.max()))
    if k != 1:
        h = len(t) // 2                    # grid.y: 0 = down walks, 1 = up walks
        print("    down walks: duration %.1f / %.1f / %.1f us, up walks: %.1f / %.1f / %.1f us (min / median / max)"
              % (d[:h].min(), np.median(d[:h]), d[:h].max(), d[h:].min(), np.median(d[h:]), d[h:].max()))
        print("    down walks by workgroup index, eighths: " + " ".join("%.1f" % d[:h][i * (h // 8):(i + 1) * (h // 8)].mean() for i in range(8)))
    order = np.argsort(b)
    q = len(t) // 8
    print("    by start order, eighths: start " + " ".join("%.1f" % b[order[i * q:(i + 1) * q]].mean() for i in range(8))
          + " | duration " + " ".join("%.1f" % d[order[i * q:(i + 1) * q]].mean() for i in range(8)))
