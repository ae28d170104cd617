import sys; sys.path.insert(0,'/root/repo')
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla, time
from functools import partial
from oracle import numpy_path as ora
from triflow_amd import Model
from triflow_amd.compilers import hip_compiler
from tests import parity_cases as pc
backend = None
if len(sys.argv) > 1 and sys.argv[1] == 'emu':
    from tests.emu.build_emu import EmuBackend
    backend = EmuBackend()
eqs = ["-dxxxx%s + k*dxx%s + %s*dx%s" % (v, v, w, v) for v, w in zip("ABCDG", "BCDGA")]
args = (eqs, list("ABCDG"), ["k"])
t=time.time()
m = Model(*args, compiler=hip_compiler if backend is None else partial(hip_compiler, backend=backend))
mo = Model(*args, compiler=ora.numpy_compiler)
N = 203
x = np.linspace(0, N*5e-2, N, endpoint=False)
rng = np.random.default_rng(0)
fd = {"x": x}
for j, k in enumerate("ABCDG"): fd[k] = 1 + 0.3*np.cos(2*np.pi*(j+1)*x/x[-1]) + 0.05*rng.standard_normal(N)
pars = dict(k=0.3, periodic=True)
F = m.F(m.fields_template(**fd), pars); Fo = mo.F(mo.fields_template(**fd), pars)
print('F equal', np.array_equal(F, Fo), 'build %.0fs' % (time.time()-t))
Jo = mo.J(mo.fields_template(**fd), pars)
n = N*5; c = 1e-4
A = sps.identity(n, format='csc') - c*Jo
rhs = rng.standard_normal(n); xs = spla.spsolve(A, rhs)
for opts in (dict(), dict(m1=8, m_upper=4)):
    s = pc.bound_solver(m, fd, pars, **opts)
    s.eval(0, with_j=True); s.factor(c)
    xx = s.solve(rhs)[0]
    print(opts, s.describe(), 'err', np.abs(xx-xs).max()/np.abs(xs).max(), s.backward_error())
