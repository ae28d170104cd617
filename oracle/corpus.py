"""Model corpus and synthetic inputs shared by the golden generator, the
oracle tests and the GPU parity tests (ORACLE / TEST INFRASTRUCTURE).

Model strings come from the reference's tests (``tests/test_model.py:21-163``,
``tests/test_simulation.py:12-17``), its README/cookbook, and the benchmark
models M1/M2/M3/M5 of SURVEY.md §8(d).
"""

import numpy as np

# name -> (equations, dependent variables, parameters, help functions)
MODELS = {
    "M1_advdiff": ("k * dxxU - c * dxU", "U", ["k", "c"], None),
    "M2_diff": ("k * dxxU", "U", "k", None),
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
    "M3_film": (["-dxq",
                 "-upwind(c, q, 2) + (h - q / h**2) / eps + We * h * dxxxh + k * dxxq",
                 "-upwind(c, T, 2) + k * dxxT - q * dxT / h"],
                ["h", "q", "T"], ["c", "eps", "We", "k"], None),
    "M5_stiff": (["Dm*dxxA - k1*A + k3*B*C",
                  "Dm*dxxB + k1*A - k3*B*C - k2*B**2",
                  "Dm*dxxC + k2*B**2 - k4*C*D",
                  "Dm*dxxD - upwind(c, D, 1) - k4*C*D",
                  "Dm*dxxE + k4*C*D"],
                 ["A", "B", "C", "D", "E"],
                 ["Dm", "k1", "k2", "k3", "k4", "c"], None),
}

DEFAULT_PARS = {
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
}


def model_args(name):
    eqs, dep, pars, helps = MODELS[name]
    return (eqs, dep, pars, helps)


def field_names(name):
    eqs, dep, pars, helps = MODELS[name]
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


# ---- the BASELINE.json configurations -------------------
def config_inputs(cfg, N=None):
    """(model name, fields dict, parameter dict, dt, scheme name) of BASELINE
    config 1, 2, 3 or 5 at ``N`` nodes (default: the configured size)."""
    two_pi = 2 * np.pi
    if cfg == 1:
        N = N or 200
        x = np.linspace(0, 1, N)
        return ("M1_advdiff", dict(x=x, U=np.cos(two_pi * x * 5)),
                dict(c=.03, k=.001, periodic=False), 0.5, "Theta")
    if cfg == 2:
        N = N or 10 ** 6
        x = np.linspace(0, 1, N, endpoint=False)
        return ("M2_diff", dict(x=x, U=np.cos(two_pi * 5 * x)),
                dict(k=1e-3, periodic=True), 1e-2, "Theta")
    if cfg == 3:
        N = N or 10 ** 6
        x = np.linspace(0, 100, N, endpoint=False)
        h = 1 + 0.1 * np.cos(two_pi * 4 * x / 100)
        return ("M3_film", dict(x=x, h=h, q=h ** 3, T=np.sin(two_pi * x / 100)),
                dict(c=1., eps=.5, We=.01, k=.05, periodic=True), 1e-3, "ROS2")
    if cfg == 5:
        N = N or 4 * 10 ** 6
        x = np.linspace(0, 1, N)
        zero = np.zeros(N)
        return ("M5_stiff",
                dict(x=x, A=np.ones(N), B=zero.copy(), C=zero.copy(),
                     D=np.exp(-((x - .5) / .1) ** 2), E=zero.copy()),
                dict(Dm=1e-4, k1=.04, k2=3e7, k3=1e4, k4=1., c=.1, periodic=False),
                1e-3, "BDF2")
    raise ValueError(cfg)


def dirichlet_hook_cfg1(t, fields, pars):
    """README hook (reference README.md:126-129)."""
    fields["U"][0] = 1
    fields["U"][-1] = 0
    return fields, pars


def dirichlet_hook_cfg5(t, fields, pars):
    fields["A"][0] = 1
    fields["A"][-1] = 1
    return fields, pars
