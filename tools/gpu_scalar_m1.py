"""Scalar models with cyclic-reduction levels: accuracy of dispersive models and speed of the
heat equation against the level-1 chunk length."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
from oracle import corpus
from tests import parity_cases as pc
from triflow_amd import Model, schemes, workloads

for name in ("kdv", "kuramoto", "burgers"):
    for periodic in (True, False):
        N = 203
        m, mo = pc.device_model(name, None), pc.oracle_model(name)
        fd = corpus.synthetic_fields(name, N, seed=7, periodic=periodic, length=N * 5e-3)
        pars = corpus.synthetic_pars(name, N, periodic)
        Jo = mo.J(mo.fields_template(**fd), pars)
        A = sps.identity(N, format="csc") - 0.01 * Jo
        rhs = np.random.default_rng(5).standard_normal(N); xs = spla.spsolve(A, rhs)
        out = []
        for m1 in (4, 8, 16, 32):
            s = pc.bound_solver(m, fd, pars, refine=0, m1=m1)
            s.eval(0, with_j=True); s.factor(0.01)
            x = s.solve(rhs)[0]
            out.append("m1=%d %s err %.1e" % (m1, s.describe()["chunks"], np.abs(x - xs).max() / np.abs(xs).max()))
        print(name, "periodic" if periodic else "clamped", " | ".join(out), flush=True)

def rate(scheme, fields, pars, dt, n):
    t = 0.0
    for _ in range(5):
        t, fields = scheme(t, fields, dt, pars)
    s = fields._device_backing().stepper.solver; s.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        t, fields = scheme(t, fields, dt, pars)
    s.sync()
    return n / (time.perf_counter() - t0)

for N in (200, 2000, 20000, 200000, 1000000):
    out = []
    for m1 in (4, 8, 16, 32):
        os.environ["TRIFLOW_M1"] = str(m1)
        name, fd, pars, dt, _ = workloads.config_inputs(2, N)
        m = Model(*workloads.model_args(name))
        out.append("m1=%d: %.0f" % (m1, rate(schemes.Theta(m), m.fields_template(**fd), pars, dt, 300)))
    print("config 2 Theta N=%-8d" % N + " | ".join(out), flush=True)
