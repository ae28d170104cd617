"""Device path against the CPU oracle over 100 time steps (SURVEY.md section 8(d):
"100-step drift reported"): relative max-norm difference of U after 1, 10, 100 steps."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import parity_cases as pc

for cfg, N, sch in [(1, 200, "Theta"), (2, 20000, "Theta"), (3, 20000, "ROS2"), (3, 20000, "RODASPR"),
                    (5, 20000, "BDF2")]:
    d = pc.drift_against_oracle(None, cfg, N, sch)
    print("config %d model, N=%d, %s: rel. difference to the oracle after 1 / 10 / 100 steps: "
          "%.1e / %.1e / %.1e" % (cfg, N, sch, d[1], d[10], d[100]), flush=True)
