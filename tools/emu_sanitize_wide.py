"""The model that made hipcc's spilling -O3 build give wrong solves (5 variables, 5-point
stencils, b = 10: tools/gpu_wide_model_check2.py), run through the SAME kernel sources
(csrc/tf_kernels.h, the host runtime) compiled for the host at -O3 with AddressSanitizer and
UndefinedBehaviorSanitizer: an uninitialised array, an out-of-range constant index or a
signed overflow in the shared source would show up here.  CPU only (the pool refuses GPU
sanitizers).  Prints the solver error next to the sanitizer verdict."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "one":
    import functools
    import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
    from oracle import numpy_path as ora
    from tests.emu import build_emu
    from tests import parity_cases as pc
    from triflow_amd import Model
    from triflow_amd.compilers import hip_compiler

    class SanBackend(build_emu.EmuBackend):
        def load(self, model, parvec_mask):
            from triflow_amd._capi import DeviceModel, Library
            so, spec = build_emu.build(model, parvec_mask, opt=sys.argv[2])
            return DeviceModel(Library(so), spec, b"")

    eqs = ["-dxxxx%s + k*dxx%s + %s*dx%s" % (v, v, w, v) for v, w in zip("ABCDG", "BCDGA")]
    args = (eqs, list("ABCDG"), ["k"])
    m = Model(*args, compiler=functools.partial(hip_compiler, backend=SanBackend()))
    mo = Model(*args, compiler=ora.numpy_compiler)
    N = 203
    x = np.linspace(0, N * 5e-2, N, endpoint=False)
    rng = np.random.default_rng(0)
    fd = {"x": x}
    for j, k in enumerate("ABCDG"):
        fd[k] = 1 + 0.3 * np.cos(2 * np.pi * (j + 1) * x / x[-1]) + 0.05 * rng.standard_normal(N)
    pars = dict(k=0.3, periodic=True)
    Jo = mo.J(mo.fields_template(**fd), pars)
    n, c = N * 5, 1e-4
    A = sps.identity(n, format="csc") - c * Jo
    rhs = rng.standard_normal(n)
    xs = spla.spsolve(A, rhs)
    for opts in (dict(), dict(m1=8, m_upper=4), dict(m1=10 ** 6)):
        s = pc.bound_solver(m, fd, pars, refine=0, **opts)
        s.eval(0, with_j=True)
        s.factor(c)
        xx = s.solve(rhs)[0]
        print(sys.argv[2], opts, s.describe()["chunks"],
              "err %.2e" % (np.abs(xx - xs).max() / np.abs(xs).max()), flush=True)
else:
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, __file__, "one", "-O3 -fsanitize=address,undefined -fno-sanitize-recover=all"],
                         env=env, capture_output=True, text=True)
    print(res.stdout, end="")
    bad = [ln for ln in res.stderr.splitlines() if "runtime error" in ln or "AddressSanitizer" in ln]
    print("sanitizer findings: %d" % len(bad), *bad[:10], sep="\n")
    print("exit code", res.returncode)
    if res.returncode:
        print(res.stderr[-3000:])
