"""First GPU contact: correctness vs oracle/scipy and kernel timings."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
from triflow_amd.model import Model
from oracle import corpus
from oracle.numpy_path import numpy_compiler

def check(cfg, N, m1=32, mu=8):
    name, fd, pars, dt, sch = corpus.config_inputs(cfg, N)
    m = Model(*corpus.model_args(name))
    cm = m._device
    periodic = pars['periodic']
    solver = cm.solver(N, periodic, 1, 0, m1=m1, m_upper=mu)
    cm.bind_inputs(solver, fd['x'], [pars[k] for k in cm.pars])
    solver.set_state(0, np.array([fd[k] for k in m._dep_vars]))
    solver.eval(0, with_j=True)
    F = solver.get_F()[0]
    out = dict(cfg=cfg, N=N, desc=solver.describe())
    if N <= 200000:
        mo = Model(*corpus.model_args(name), compiler=numpy_compiler)
        fo = mo.fields_template(**fd)
        Fo = mo.F(fo, pars); Jo = mo.J(fo, pars)
        J = cm.pattern(N, periodic).assemble(solver.get_J()[0])
        out['F_bitexact'] = bool(np.array_equal(F, Fo)); out['F_maxdiff'] = float(np.abs(F-Fo).max())
        out['J_maxdiff'] = float(abs(J-Jo).max())
        c = 0.2928932188134*dt
        solver.factor(c)
        rhs = dt*Fo
        x = solver.solve(rhs)[0]
        A = sps.identity(N*m._nvar, format='csc') - c*Jo
        xs = spla.spsolve(A, rhs)
        out['solve_relerr_vs_superlu'] = float(np.abs(x-xs).max()/np.abs(xs).max())
        out['resid_gpu'] = float(np.abs(A@x-rhs).max()/np.abs(rhs).max()); out['resid_superlu'] = float(np.abs(A@xs-rhs).max()/np.abs(rhs).max())
    # timing
    solver.timing(True); solver.timing_reset()
    tab = __import__('triflow_amd.tableaux', fromlist=['TABLEAUX']).TABLEAUX['ROS2']
    t0 = time.perf_counter()
    nst = 10
    for i in range(nst):
        solver.step_row(i % 2, (i+1) % 2, dt, tab.alpha, tab.gamma, tab.b, None, True, want_err=False)
    solver.sync()
    wall = time.perf_counter() - t0
    rep = solver.timing_report()
    out['ros2_ms_per_step_wall_with_events'] = wall/nst*1e3
    out['kernels_ms_per_launch'] = {k: round(v[0]/v[1], 4) for k, v in rep.items()}
    out['kernels_ms_per_step'] = {k: round(v[0]/nst, 4) for k, v in rep.items()}
    solver.timing(False)
    t0 = time.perf_counter()
    for i in range(nst):
        solver.step_row(i % 2, (i+1) % 2, dt, tab.alpha, tab.gamma, tab.b, None, True, want_err=False)
    solver.sync()
    out['ros2_ms_per_step_wall'] = (time.perf_counter()-t0)/nst*1e3
    U = solver.get_state(nst % 2)
    out['finite'] = bool(np.isfinite(U).all())
    print(json.dumps(out, indent=1)); sys.stdout.flush()
    solver.close()

if __name__ == '__main__':
    check(2, 20000, 8, 3)
    check(3, 20000, 8, 3)
    check(5, 20000, 8, 3)
    check(3, 200000)
    check(2, 10**6)
    check(3, 10**6)
    check(5, 4*10**6)
