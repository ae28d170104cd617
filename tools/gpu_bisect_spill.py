"""Which kernel of the spilling -O3 build of the b = 10 model (tools/gpu_wide_model_check2.py)
gives the wrong solve?  Builds the model at -O3 (forced: the -O1 fallback of
compilers.build_code_object is bypassed) and at -O1, then runs the factor/solve check with
every solver kernel taken, one at a time, from the -O1 code object (TF_ALT_HSACO / TF_ALT_MASK
hook of tf_backend_hip.cpp).  A kernel whose swap alone repairs the result is the miscompiled
one; prints its resource usage in both builds.  GPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EQS = ["-dxxxx%s + k*dxx%s + %s*dx%s" % (v, v, w, v) for v, w in zip("ABCDG", "BCDGA")]
ARGS = (EQS, list("ABCDG"), ["k"])


def build(opt):
    """hsaco of the model at the given -O level, without the spill fallback."""
    from triflow_amd import Model, codegen, compilers
    m = Model(*ARGS, hold_compilation=True)
    body, spec = codegen.lower_model(m, parvec_mask=0, seg=8, sweep_block=256)
    src = compilers._TU_HEAD % "" + body + compilers._TU_TAIL
    out = os.path.join(ROOT, "gpurun_out", "bisect")
    os.makedirs(out, exist_ok=True)
    hip = os.path.join(out, "wide_%s.hip" % opt.strip("-"))
    with open(hip, "w") as f:
        f.write(src)
    hsaco = hip[:-4] + ".hsaco"
    flags = [opt] + [f for f in compilers.HIPCC_FLAGS if not f.startswith("-O")]
    res = subprocess.run([compilers._hipcc(), *flags, "-I", compilers.CSRC, "--genco", "--no-gpu-bundle-output",
                          "-Rpass-analysis=kernel-resource-usage", "-o", hsaco, hip], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    return hsaco, compilers._parse_resource_usage(res.stderr)


if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
    from oracle import numpy_path as ora
    from tests import parity_cases as pc
    from triflow_amd import Model
    m, mo = Model(*ARGS), Model(*ARGS, compiler=ora.numpy_compiler)
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
    errs = []
    for opts in (dict(), dict(m1=8, m_upper=4), dict(m1=10 ** 6)):      # the plans of gpu_wide_model_check2.py
        s = pc.bound_solver(m, fd, pars, refine=0, **opts)
        s.eval(0, with_j=True)
        s.factor(c)
        xx = s.solve(rhs)[0]
        errs.append("%.2e" % (np.abs(xx - xs).max() / np.abs(xs).max()))
    print(" ".join(errs))
else:
    from triflow_amd import compilers
    h3, u3 = build("-O3")
    h1, u1 = build("-O1")
    lib = compilers.HipBackend().library()
    names = lib.kernel_names()
    spilled = [k for k, u in u3.items() if u.get("ScratchSize", 0) > 0]
    print("hipcc:", compilers.hipcc_version())
    print("kernels with scratch at -O3:", {k: u3[k] for k in spilled})

    def run(env_extra):
        # the product path would rebuild a spilling model at -O1: force -O3 through the alt hook
        env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
        r = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
        return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "failed: " + r.stderr[-300:]
    allmask = (1 << len(names)) - 1
    print("all kernels from -O1 (product default for this model):", run({}))
    print("all kernels from -O3:", run({"TF_ALT_HSACO": h3, "TF_ALT_MASK": str(allmask)}))
    solver = [i for i, n in enumerate(names) if n.startswith(("tfk_l1_", "tfk_bt_", "tfk_top_", "tfk_cr_"))]
    for i in solver:
        # everything from -O3 except kernel i (which stays the -O1 product build)
        err = run({"TF_ALT_HSACO": h3, "TF_ALT_MASK": str(allmask & ~(1 << i))})
        print("%-20s from -O1, rest -O3: err %s   -O3 usage %s" % (names[i], err, u3.get(names[i])))
