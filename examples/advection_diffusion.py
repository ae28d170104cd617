"""The example of the reference's README (README.md:104-139; plotting left out,
``Simulation`` called with the keyword names of ``simulation.py:160-175``), only the import
line differs: advection-diffusion of a cosine with Dirichlet values set by a Python hook."""
import numpy as np
from triflow_amd import Model, Simulation

model = Model("k * dxxU - c * dxU", "U", ["k", "c"])

x, dx = np.linspace(0, 1, 200, retstep=True)
U = np.cos(2 * np.pi * x * 5)
fields = model.fields_template(x=x, U=U)
parameters = dict(c=.03, k=.001, dx=dx, periodic=False)


def dirichlet_condition(t, fields, pars):
    fields.U[0] = 1
    fields.U[-1] = 0
    return fields, pars


simul = Simulation(model, fields, parameters, dt=5E-1, tmax=2.5, hook=dirichlet_condition)
for i, (t, fields) in enumerate(simul):
    print("iteration: %d  t: %g  U[1:4] = %s" % (i, t, np.asarray(fields.U)[1:4]))
