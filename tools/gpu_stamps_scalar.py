"""In-kernel phase timing of the two-launch solve of the scalar models (diagnostic build -DTF_STAMPS):
shader-clock stamps of the middle workgroup of tfk_s_fwd / tfk_s_bwd and of the workgroup that arrives last."""
import os, sys
os.environ["TRIFLOW_HIPCC_EXTRA"] = (os.environ.get("TRIFLOW_HIPCC_EXTRA", "") + " -DTF_STAMPS").strip()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from triflow_amd import Model, schemes, workloads

name, fd, pars, dt, _ = workloads.config_inputs(2, None)
model = Model(*workloads.model_args(name))
sch, f, t = schemes.Theta(model), model.fields_template(**fd), 0.0
for _ in range(3):
    t, f = sch(t, f, dt, pars)
s = f._device_backing().stepper.solver
s.debug_stamps()
for _ in range(5):
    t, f = sch(t, f, dt, pars)
s.sync()
st = s.debug_stamps().astype(np.int64)
print("levels", s.describe()["chunks"])
r = st[1]
print("tfk_s_fwd, middle workgroup: stage requests %d | walks %d | barrier %d | level 2 %d | count + barrier %d cycles"
      % (r[58] - r[57], r[59] - r[58], r[60] - r[59], r[61] - r[60], r[62] - r[61]))
print("           last arriver: level 3 %d cycles; its start %.2f us and end %.2f us after the middle workgroup's entry"
      % (r[56] - r[55], (r[32] - r[30]) / 100.0, (r[31] - r[30]) / 100.0))
print("tfk_s_bwd, middle workgroup: level 2 %d | barrier %d | level-1 back-substitution %d cycles"
      % (r[51] - r[50], r[52] - r[51], r[53] - r[52]))
