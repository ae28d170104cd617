"""Model corpus and synthetic inputs shared by the golden generator, the
oracle tests and the GPU parity tests (ORACLE / TEST INFRASTRUCTURE).

Model strings come from the reference's tests (``tests/test_model.py:21-163``,
``tests/test_simulation.py:12-17``), its README/cookbook, and the benchmark
models M1/M2/M3/M5 of SURVEY.md §8(d).
"""

import numpy as np

from triflow_amd.workloads import BENCH_MODELS

# name -> (equations, dependent variables, parameters, help functions)
MODELS = {
    "M1_advdiff": BENCH_MODELS["M1_advdiff"],
    "M2_diff": BENCH_MODELS["M2_diff"],
    "diff_nested": ("k * dx(dxU)", "U", "k", None),
    "diff_list": (["k * dxxU"], ["U"], ["k"], None),
    "heat_nopar": ("dxxU", "U", None, None),
    "bivar": (["k1 * dxx(v)", "k2 * dxx(u)"], ["u", "v"], ["k1", "k2"], None),
    "helper": (["k * dxxU + s"], "U", "k", "s"),
    "helper_d": (["k * dxxU + dxs * U"], "U", "k", "s"),
    "upwind1_const": (["upwind(1, U, 1)"], "U", "k", "s"),
    "upwind2_const": (["upwind(1, U, 2)"], "U", "k", "s"),
    "upwind3_const": (["upwind(1, U, 3)"], "U", "k", "s"),
    "upwind1_par": ("-upwind(c, U, 1) + k * dxxU", "U", ["c", "k"], None),
    "upwind2_par": ("-upwind(c, U, 2) + k * dxxU", "U", ["c", "k"], None),
    "upwind3_par": ("-upwind(c, U, 3) + k * dxxU", "U", ["c", "k"], None),
    "upwind2_state": (["upwind(U, U, 2)"], "U", "k", "s"),
    "burgers": ("k * dxxU - U * dxU", "U", "k", None),
    "kdv": ("-6 * U * dxU - dxxxU", "U", None, None),
    "kuramoto": ("-dxxxxU - dxxU - U * dxU", "U", None, None),
    "wave": (["dxV", "c**2 * dxU"], ["U", "V"], "c", None),
    "nonlin": ("k * dxxU + exp(-U**2) - sqrt(1 + U**2) + U**3", "U", "k", None),
    "M3_film": BENCH_MODELS["M3_film"],
    "M5_stiff": BENCH_MODELS["M5_stiff"],
    # wide blocks: 4 variables with 5-point stencils (b = 8), 6 variables (b = 6)
    "wide4": (["-dxxxxA - dxxA + B*dxA", "k*dxxB - A*dxxxC", "k*dxxC + dxD*A",
               "k*dxxD - dxxxxD + B"], ["A", "B", "C", "D"], "k", None),
    "six": (["k*dxx%s + %s*dx%s" % (v, w, v) for v, w in zip("ABCDGH", "BCDGHA")],
            list("ABCDGH"), "k", None),
    # a bare ``E`` is SymPy's Euler number in the reference's sympify namespace
    "euler_const": ("k * dxxU + E * U + pi * dxU", "U", "k", None),
}

#: models without reference goldens: only used to exercise the banded solver at block sizes the
#: corpus above does not reach (b = mp * nvar = 3, 4, 7), against SuperLU on the oracle's Jacobian
SOLVER_MODELS = {
    "tri3": (["k*dxx%s + %s*dx%s - %s" % (v, w, v, v) for v, w in zip("ABC", "BCA")], list("ABC"), "k", None),
    "quad4": (["k*dxx%s + %s*dx%s" % (v, w, v) for v, w in zip("ABCD", "BCDA")], list("ABCD"), "k", None),
    "pair4": (["-dxxxxA + k*dxxA + B*dxA", "k*dxxB - A*dxxxB"], ["A", "B"], "k", None),
    "seven": (["k*dxx%s + %s*dx%s" % (v, w, v) for v, w in zip("ABCDGHK", "BCDGHKA")], list("ABCDGHK"), "k", None),
}
MODELS_ALL = dict(MODELS, **SOLVER_MODELS)

DEFAULT_PARS = {
    "tri3": dict(k=.2), "quad4": dict(k=.2), "pair4": dict(k=.3), "seven": dict(k=.2),
    "M1_advdiff": dict(k=.001, c=.03),
    "M2_diff": dict(k=1e-3), "diff_nested": dict(k=1e-3), "diff_list": dict(k=1e-3),
    "heat_nopar": {},
    "bivar": dict(k1=1., k2=.7),
    "helper": dict(k=.5), "helper_d": dict(k=.5),
    "upwind1_const": dict(k=1.), "upwind2_const": dict(k=1.), "upwind3_const": dict(k=1.),
    "upwind1_par": dict(c=.8, k=.01), "upwind2_par": dict(c=-.8, k=.01),
    "upwind3_par": dict(c=.8, k=.01),
    "upwind2_state": dict(k=1.),
    "burgers": dict(k=.05), "kdv": {}, "kuramoto": {},
    "wave": dict(c=2.),
    "nonlin": dict(k=.1),
    "M3_film": dict(c=1., eps=.5, We=.01, k=.05),
    "M5_stiff": dict(Dm=1e-4, k1=.04, k2=3e7, k3=1e4, k4=1., c=.1),
    "wide4": dict(k=.3), "six": dict(k=.2), "euler_const": dict(k=.1),
}


def model_args(name):
    eqs, dep, pars, helps = MODELS_ALL[name]
    return (eqs, dep, pars, helps)


def field_names(name):
    eqs, dep, pars, helps = MODELS_ALL[name]
    as_list = lambda a: [] if a is None else ([a] if isinstance(a, str) else list(a))
    return as_list(dep), as_list(helps), as_list(pars)


def synthetic_fields(name, N, seed=0, length=10.0, periodic=True):
    """Smooth, strictly positive-where-needed fields plus a seeded
    perturbation (so symmetric cancellations cannot hide indexing bugs)."""
    dep, helps, _ = field_names(name)
    rng = np.random.default_rng(seed)
    x = np.linspace(0, length, N, endpoint=not periodic)
    out = {"x": x}
    for j, key in enumerate(dep + helps):
        base = 1.0 + 0.3 * np.cos(2 * np.pi * (j + 1) * x / length + 0.4 * j)
        out[key] = base + 0.05 * rng.standard_normal(N)
    if name == "M5_stiff":
        for key in ("B", "C", "E"):
            out[key] = 1e-4 * np.abs(out[key])
    return out


def synthetic_pars(name, N, periodic, per_node=False, seed=1):
    pars = dict(DEFAULT_PARS[name])
    if per_node:
        rng = np.random.default_rng(seed)
        for key in pars:
            pars[key] = pars[key] * (1.0 + 0.1 * rng.random(N))
    pars["periodic"] = periodic
    return pars


from triflow_amd.workloads import config_inputs  # noqa: E402,F401  (BASELINE configurations)


def dirichlet_hook_cfg1(t, fields, pars):
    """README hook (reference README.md:126-129)."""
    fields["U"][0] = 1
    fields["U"][-1] = 0
    return fields, pars


def dirichlet_hook_cfg5(t, fields, pars):
    fields["A"][0] = 1
    fields["A"][-1] = 1
    return fields, pars
