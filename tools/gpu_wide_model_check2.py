"""b = 10 model (5 variables, 5-point stencils): raw backward error of the factorisation
without refinement, for the default and an -O1 build (spilling level-1 kernels)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
    from oracle import numpy_path as ora
    from triflow_amd import Model
    from tests import parity_cases as pc
    eqs = ["-dxxxx%s + k*dxx%s + %s*dx%s" % (v, v, w, v) for v, w in zip("ABCDG", "BCDGA")]
    args = (eqs, list("ABCDG"), ["k"])
    m, mo = Model(*args), Model(*args, compiler=ora.numpy_compiler)
    N = 203
    x = np.linspace(0, N * 5e-2, N, endpoint=False)
    rng = np.random.default_rng(0)
    fd = {"x": x}
    for j, k in enumerate("ABCDG"):
        fd[k] = 1 + 0.3 * np.cos(2 * np.pi * (j + 1) * x / x[-1]) + 0.05 * rng.standard_normal(N)
    pars = dict(k=0.3, periodic=True)
    Jo = mo.J(mo.fields_template(**fd), pars)
    n = N * 5; c = 1e-4
    A = sps.identity(n, format="csc") - c * Jo
    rhs = rng.standard_normal(n); xs = spla.spsolve(A, rhs)
    for opts in (dict(), dict(m1=8, m_upper=4), dict(m1=10 ** 6)):
        s = pc.bound_solver(m, fd, pars, refine=0, **opts)
        s.eval(0, with_j=True); s.factor(c)
        xx = s.solve(rhs)[0]
        r = np.abs(rhs - A @ xx).max() / np.abs(rhs).max()
        print(os.environ.get("TRIFLOW_HIPCC_OPT", "-O3"), opts, s.describe()["chunks"], "err %.2e resid %.2e" % (np.abs(xx - xs).max() / np.abs(xs).max(), r), flush=True)
else:
    for opt in ("-O3", "-O1"):
        subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, TRIFLOW_HIPCC_OPT=opt))
