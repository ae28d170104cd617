import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import corpus
from triflow_amd.model import Model
from triflow_amd.tableaux import TABLEAUX
name, fd, pars, dt, _ = corpus.config_inputs(3)
m = Model(*corpus.model_args(name)); cm = m._device
tab = TABLEAUX['ROS2']
ref = None
for opts in [dict(tail_chunks=-1), dict(), dict(m_upper=4), dict(m_upper=4, tail_chunks=512), dict(m_upper=3), dict(m_upper=6), dict(m_upper=8, tail_chunks=512), dict(m1=24), dict(m1=48), dict(m1=64, m_upper=4)]:
    s = cm.solver(fd['x'].size, True, 1, 0, **opts)
    cm.bind_inputs(s, fd['x'], [pars[k] for k in cm.pars])
    s.set_state(0, np.array([fd[k] for k in m._dep_vars]))
    for i in range(3): s.step_row(i%2, (i+1)%2, dt, tab.alpha, tab.gamma, tab.b, None, True, want_err=False)
    s.sync()
    U = s.get_state(1)
    if ref is None: ref = U
    t0 = time.perf_counter()
    for i in range(20): s.step_row((i+1)%2, i%2, dt, tab.alpha, tab.gamma, tab.b, None, True, want_err=False)
    s.sync(); el = (time.perf_counter()-t0)/20
    print(opts, s.describe()['chunks'], 'omega', s.backward_error(), 'maxdiff vs first %.2e' % np.abs(U-ref).max(), 'ms/step %.3f' % (el*1e3)); sys.stdout.flush()
    s.close()
