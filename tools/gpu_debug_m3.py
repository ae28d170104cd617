import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
from triflow_amd.model import Model
from oracle import corpus
from oracle.numpy_path import numpy_compiler

def run(name, N, cfgs, c=3e-4, periodic=True, cfg3=False):
    if cfg3:
        _, fd, pars, dt, _ = corpus.config_inputs(3, N)
    else:
        fd = corpus.synthetic_fields(name, N, seed=3, periodic=periodic, length=N*5e-3)
        pars = corpus.synthetic_pars(name, N, periodic)
    m = Model(*corpus.model_args(name))
    mo = Model(*corpus.model_args(name), compiler=numpy_compiler)
    fo = mo.fields_template(**fd); Jo = mo.J(fo, pars)
    cm = m._device
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal(N*m._nvar)
    A = sps.identity(N*m._nvar, format='csc') - c*Jo
    xs = spla.spsolve(A, rhs)
    for (m1, mu) in cfgs:
        solver = cm.solver(N, pars['periodic'], 1, 0, m1=m1, m_upper=mu)
        cm.bind_inputs(solver, fd['x'], [pars[k] for k in cm.pars], [fd[k] for k in m._help_funcs] if cm.nh else None)
        solver.set_state(0, np.array([fd[k] for k in m._dep_vars]))
        solver.eval(0, with_j=True)
        solver.factor(c)
        x = solver.solve(rhs)[0]
        print(name, N, solver.describe()['chunks'], 'err %.2e' % (np.abs(x-xs).max()/np.abs(xs).max()),
              'resid %.2e (superlu %.2e)' % (np.abs(A@x-rhs).max()/np.abs(rhs).max(), np.abs(A@xs-rhs).max()/np.abs(rhs).max()))
        sys.stdout.flush()

cfgs = [(10**7, 8), (8, 3), (32, 8)]
run('kdv', 2000, cfgs)
run('kuramoto', 2000, cfgs)
run('upwind2_par', 2000, cfgs)
run('M3_film', 2000, cfgs, cfg3=True)
run('M3_film', 2000, cfgs)
run('M5_stiff', 2000, cfgs)
run('bivar', 2000, cfgs)
