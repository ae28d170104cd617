"""Generate ``tests/golden/*`` by running the REFERENCE itself.

ORACLE TOOLING.  Run in the build container only (``python oracle/gen_golden.py``):
imports the reference's hot-path modules from ``/root/reference`` through
``oracle/ref_loader.py`` and stores inputs + outputs as small fixtures.  The
fixtures are data (arrays and strings), never reference source.

Written files
  ir.json            G4  symbolic IR of every corpus model
  fj_<model>.npz     G1  F and J (CSC triplets) for periodic/clamped,
                         scalar/per-node parameters
  steps.npz          G2  U after 1..5 steps for every scheme, fixed and
                         adaptive, with and without the README Dirichlet hook
  simulation.npz     G3  ``Simulation`` output sequence of config 1 with
                         ``time_stepping`` True / False
  vode_bdf.npz       G5  scipy_ode(vode, bdf) trajectory (BDF-2 sanity anchor)
  versions.json      library versions the vectors are tied to
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import corpus, ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_FJ = 24


def ref_model(ref, name, **kw):
    eqs, dep, pars, helps = corpus.model_args(name)
    return ref_loader.numpy_model(ref, eqs, dep, pars, helps, **kw)


def gen_ir(ref):
    out = {}
    for name in corpus.MODELS:
        m = ref_model(ref, name)
        out[name] = dict(
            args=m._args,
            F=[str(e) for e in m.F_array.tolist()],
            sparse_indices=[int(i) for i in m._sparse_indices[0]],
            J=[str(e) for e in m._J_sparse_array.tolist()],
            bounds=list(m._bounds), window=int(m._window_range), nvar=int(m._nvar))
    for name, kw in (("heat_nopar", dict(simplify=True)), ("heat_nopar", dict(fdiff_jac=True)),
                     ("burgers", dict(fdiff_jac=True))):
        m = ref_model(ref, name, **kw)
        key = name + "|" + ",".join(sorted(kw))
        out[key] = dict(F=[str(e) for e in m.F_array.tolist()],
                        sparse_indices=[int(i) for i in m._sparse_indices[0]],
                        J=[str(e) for e in m._J_sparse_array.tolist()])
    with open(os.path.join(OUT, "ir.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def gen_fj(ref):
    for name in corpus.MODELS:
        m = ref_model(ref, name)
        store = {}
        for periodic in (True, False):
            for per_node in (False, True):
                if per_node and not corpus.DEFAULT_PARS[name]:
                    continue
                tag = "%s_%s" % ("per" if periodic else "clamp",
                                 "vec" if per_node else "sca")
                fdict = corpus.synthetic_fields(name, N_FJ, seed=3, periodic=periodic)
                pars = corpus.synthetic_pars(name, N_FJ, periodic, per_node)
                fields = m.fields_template(**fdict)
                F = m.F(fields, pars)
                J = m.J(fields, pars)
                J.sum_duplicates()
                J.sort_indices()
                store[tag + "_F"] = F
                store[tag + "_Jdata"] = J.data
                store[tag + "_Jindices"] = J.indices.astype(np.int64)
                store[tag + "_Jindptr"] = J.indptr.astype(np.int64)
        np.savez_compressed(os.path.join(OUT, "fj_%s.npz" % name), **store)


STEP_CASES = [
    # (case name, model, N, periodic, dt, hook name)
    ("cfg1", "M1_advdiff", 40, False, 0.5, "cfg1"),
    ("cfg1_nohook", "M1_advdiff", 40, False, 0.5, None),
    ("diff_per", "M2_diff", 32, True, 1e-2, None),
    ("film_per", "M3_film", 32, True, 1e-3, None),
    ("film_clamp", "M3_film", 32, False, 1e-3, None),
    ("stiff_clamp", "M5_stiff", 24, False, 1e-3, "cfg5"),
    ("burgers_per", "burgers", 32, True, 5e-3, None),
]
HOOKS = {"cfg1": corpus.dirichlet_hook_cfg1, "cfg5": corpus.dirichlet_hook_cfg5, None: None}


def step_inputs(case):
    cname, mname, N, periodic, dt, hook = case
    if cname.startswith("cfg1"):
        _, fdict, pars, _, _ = corpus.config_inputs(1, N)
    elif cname.startswith("film"):
        # benchmark IC of config 3 on a short domain with the benchmark dx
        x = np.linspace(0, 3.2, N, endpoint=not periodic)
        h = 1 + 0.1 * np.cos(2 * np.pi * x / 3.2)
        fdict = dict(x=x, h=h, q=h ** 3, T=np.sin(2 * np.pi * x / 3.2))
        pars = dict(c=1., eps=.5, We=.01, k=.05, periodic=periodic)
    elif cname.startswith("stiff"):
        _, fdict, pars, _, _ = corpus.config_inputs(5, N)
    else:
        fdict = corpus.synthetic_fields(mname, N, seed=5, periodic=periodic)
        pars = corpus.synthetic_pars(mname, N, periodic)
    return fdict, pars


def gen_steps(ref):
    S = ref.schemes
    store = {}
    schemes = {
        "Theta1": lambda m: S.Theta(m, theta=1),
        "Theta05": lambda m: S.Theta(m, theta=0.5),
        "Theta0": lambda m: S.Theta(m, theta=0),
        "ROS2": lambda m: S.ROS2(m),
        "ROS3PRw": lambda m: S.ROS3PRw(m, time_stepping=False),
        "ROS3PRL": lambda m: S.ROS3PRL(m, time_stepping=False),
        "RODASPR": lambda m: S.RODASPR(m, time_stepping=False),
        "ROS3PRw_adapt": lambda m: S.ROS3PRw(m, tol=1e-1),
        "ROS3PRL_adapt": lambda m: S.ROS3PRL(m, tol=1e-1),
        "RODASPR_adapt": lambda m: S.RODASPR(m, tol=1e-1),
    }
    for case in STEP_CASES:
        cname, mname, N, periodic, dt, hook = case
        m = ref_model(ref, mname)
        fdict, pars = step_inputs(case)
        for sname, make in schemes.items():
            if sname == "Theta0" and cname not in ("cfg1", "diff_per"):
                continue     # forward Euler is unstable on the stiff cases
            scheme = make(m)
            fields = m.fields_template(**fdict)
            t = 0.0
            traj = []
            kw = dict(hook=HOOKS[hook]) if hook else {}
            with np.errstate(all="ignore"):
                for _ in range(5):
                    t, fields = scheme(t, fields, dt, pars, **kw)
                    traj.append(fields.uflat.copy())
            store["%s|%s" % (cname, sname)] = np.array(traj)
    np.savez_compressed(os.path.join(OUT, "steps.npz"), **store)


def gen_simulation(ref):
    S = ref.schemes
    store = {}
    for ts in (True, False):
        m = ref_model(ref, "M1_advdiff")
        _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
        sim = ref.Simulation(m, fdict, pars, dt, hook=corpus.dirichlet_hook_cfg1,
                             tmax=2.5, scheme=S.Theta, time_stepping=ts)
        ts_list, us = [], []
        for t, fields in sim:
            ts_list.append(t)
            us.append(fields.uflat.copy())
        store["Theta_ts%i_t" % ts] = np.array(ts_list)
        store["Theta_ts%i_U" % ts] = np.array(us)
    # default scheme (RODASPR, adaptive) on the README problem
    m = ref_model(ref, "M1_advdiff")
    _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
    sim = ref.Simulation(m, fdict, pars, dt, hook=corpus.dirichlet_hook_cfg1, tmax=2.5)
    us = [fields.uflat.copy() for t, fields in sim]
    store["RODASPR_default_U"] = np.array(us)
    np.savez_compressed(os.path.join(OUT, "simulation.npz"), **store)


def gen_vode(ref):
    S = ref.schemes
    m = ref_model(ref, "M2_diff")
    x = np.linspace(0, 10, 50, endpoint=False)
    fields = m.fields_template(x=x, U=np.cos(x * 2 * np.pi / 10))
    pars = dict(periodic=True, k=1)
    scheme = S.scipy_ode(m, integrator="vode", method="bdf", rtol=1e-10, atol=1e-12)
    t, traj = 0.0, []
    for _ in range(10):
        t, fields = scheme(t, fields, 0.1, pars)
        traj.append(fields.uflat.copy())
    np.savez_compressed(os.path.join(OUT, "vode_bdf.npz"), U=np.array(traj), x=x)


def main():
    import scipy
    import sympy
    os.makedirs(OUT, exist_ok=True)
    ref = ref_loader.load()
    gen_ir(ref)
    gen_fj(ref)
    gen_steps(ref)
    gen_simulation(ref)
    gen_vode(ref)
    with open(os.path.join(OUT, "versions.json"), "w") as f:
        json.dump(dict(numpy=np.__version__, scipy=scipy.__version__,
                       sympy=sympy.__version__, python=sys.version.split()[0]), f)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
