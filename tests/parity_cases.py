"""Parity checks shared by the CPU emulation suite (tests/test_emu_parity.py,
kernel bodies executed on the host) and the GPU suite (tests/test_gpu_parity.py,
the real HIP path through libtriflow_hip.so).  Every check compares the device
path with the oracle / the reference's golden vectors on the same inputs."""
import os
from functools import partial

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from oracle import corpus, numpy_path as ora
from oracle.gen_golden import HOOKS, N_FJ, STEP_CASES, step_inputs
from triflow_amd import Model, Simulation, schemes
from triflow_amd.compilers import hip_compiler
from triflow_amd.device import DirichletHook

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

#: models whose expressions only use + - * / sqrt and integer powers: F and J
#: must be bit-identical to the NumPy path.  The others call exp/... from a
#: different libm: 4 ulp of the largest term.
TRANSCENDENTAL = {"nonlin"}


def device_model(name, backend, **kw):
    eqs, dep, pars, helps = corpus.model_args(name)
    compiler = hip_compiler if backend is None else partial(hip_compiler, backend=backend)
    return Model(eqs, dep, pars, helps, compiler=compiler, **kw)


def oracle_model(name, **kw):
    eqs, dep, pars, helps = corpus.model_args(name)
    return Model(eqs, dep, pars, helps, compiler=ora.numpy_compiler, **kw)


# ---------------------------------------------------------------- seam #1: F / J
def check_FJ_golden(name, backend):
    """G1 vectors of the reference: F and the CSC Jacobian."""
    g = np.load(os.path.join(GOLDEN, "fj_%s.npz" % name))
    m = device_model(name, backend)
    for periodic in (True, False):
        for per_node in (False, True):
            if per_node and not corpus.DEFAULT_PARS[name]:
                continue
            tag = "%s_%s" % ("per" if periodic else "clamp", "vec" if per_node else "sca")
            fd = corpus.synthetic_fields(name, N_FJ, seed=3, periodic=periodic)
            pars = corpus.synthetic_pars(name, N_FJ, periodic, per_node)
            fields = m.fields_template(**fd)
            F = m.F(fields, pars)
            J = m.J(fields, pars)
            assert isinstance(J, sps.csc_matrix)
            assert np.array_equal(J.indptr, g[tag + "_Jindptr"])
            assert np.array_equal(J.indices, g[tag + "_Jindices"])
            if name in TRANSCENDENTAL:
                assert np.allclose(F, g[tag + "_F"], rtol=1e-14, atol=1e-14 * np.abs(F).max())
                assert np.allclose(J.data, g[tag + "_Jdata"], rtol=1e-14,
                                   atol=1e-14 * np.abs(J.data).max())
            else:
                assert np.array_equal(F, g[tag + "_F"]), (name, tag)
                # three clamped ghost columns folding onto one boundary column are
                # summed in a different order than SciPy's COO->CSC (1 ulp)
                d = np.abs(J.data - g[tag + "_Jdata"])
                assert (d <= 2 * np.spacing(np.abs(g[tag + "_Jdata"]))).all(), (name, tag)
                assert (d == 0).mean() > 0.97
            dense = m.J(fields, pars, sparse=False)
            assert np.array_equal(np.asarray(dense), J.toarray())


def check_FJ_bitexact_large(name, backend, N):
    """Against the oracle at a size with ragged chunks and several sweep segments."""
    m, mo = device_model(name, backend), oracle_model(name)
    for periodic in (True, False):
        fd = corpus.synthetic_fields(name, N, seed=11, periodic=periodic, length=N * 1e-2)
        pars = corpus.synthetic_pars(name, N, periodic)
        F = m.F(m.fields_template(**fd), pars)
        Fo = mo.F(mo.fields_template(**fd), pars)
        assert np.array_equal(F, Fo)
        J = m.J(m.fields_template(**fd), pars)
        Jo = mo.J(mo.fields_template(**fd), pars)
        assert abs(J - Jo).max() <= 4 * np.spacing(abs(Jo).max())


# ---------------------------------------------------------------- seam #3: solver
def check_tiny_grids(backend):
    """Grids shorter than one stencil window (N < 2*mp + 1): the reference pads them with ghost cells
    that wrap or clamp onto nodes already in the window (compilers.py:257-264) and hands SuperLU the small
    dense system.  F bit for bit, J to the order of the duplicate sums, one Theta and one ROS2 step
    (the dense path tfk_tiny_*) against the oracle."""
    cases = [("M1_advdiff", 2, True), ("M1_advdiff", 2, False), ("M1_advdiff", 1, False),
             ("kdv", 2, True), ("kdv", 3, True), ("kdv", 4, False), ("kdv", 3, False),
             ("M3_film", 3, True), ("M3_film", 4, True), ("M3_film", 4, False), ("M3_film", 2, False)]
    for name, N, periodic in cases:
        m, mo = device_model(name, backend), oracle_model(name)
        fd = corpus.synthetic_fields(name, max(N, 2), seed=4, periodic=periodic, length=0.5 * max(N, 2))
        if N == 1:
            # (dx from a one-node grid is 0/0 in the reference, compilers.py:234-237: skip the degenerate spacing)
            continue
        pars = corpus.synthetic_pars(name, N, periodic)
        F, Fo = m.F(m.fields_template(**fd), pars), mo.F(mo.fields_template(**fd), pars)
        assert np.array_equal(F, Fo), (name, N, periodic)
        J, Jo = m.J(m.fields_template(**fd), pars), mo.J(mo.fields_template(**fd), pars)
        assert np.array_equal(J.indptr, Jo.indptr) and np.array_equal(J.indices, Jo.indices)
        assert (np.abs(J.data - Jo.data) <= 4 * np.spacing(np.abs(Jo.data).max())).all(), (name, N, periodic)
        for sd, so in ((schemes.Theta(m), ora.Theta(mo)), (schemes.ROS2(m), ora.ROS2(mo))):
            _, fdv = sd(0.0, m.fields_template(**fd), 1e-2, pars)
            _, fov = so(0.0, mo.fields_template(**fd), 1e-2, pars)
            err = np.abs(fdv.uflat - fov.uflat).max() / np.abs(fov.uflat).max()
            assert err <= 1e-11, (name, N, periodic, type(sd).__name__, err)


def check_proportional_entries(backend):
    """Jacobian entries that the generated code marks as a power-of-two multiple of another entry
    (codegen._proportional_entries: the solver walks, J @ v and the monitors load one and scale)
    are that multiple *bit for bit* in the value table the sweep writes -- per-node parameters
    included -- and the models of BASELINE have some (film: 3, stiff: 3)."""
    found = {}
    for name in ("M3_film", "M5_stiff", "wide4", "burgers", "M1_advdiff"):
        m = device_model(name, backend)
        for per_node in (False, True):
            fd = corpus.synthetic_fields(name, 301, seed=9, periodic=True)
            pars = corpus.synthetic_pars(name, 301, True, per_node)
            cm = m._device
            solver = cm.solver(301, True, 1, cm.parvec_mask_of([pars[k] for k in cm.pars]))
            cm.bind_inputs(solver, fd["x"], [pars[k] for k in cm.pars],
                           [fd[k] for k in m._help_funcs] if cm.nh else None)
            solver.set_state(0, np.array([fd[k] for k in m._dep_vars]))
            solver.eval(0, with_j=True)
            table = solver.get_J()[0]                              # [N][nnz]
            spec = solver.model.spec
            pairs = [(k, a, sc) for k, (a, sc) in enumerate(zip(spec["j_alias"], spec["j_alias_scale"])) if a >= 0]
            if not per_node:
                found[name] = len(pairs)
            for k, a, sc in pairs:
                assert not spec["j_uniform"][k] and not spec["j_uniform"][a] and spec["j_alias"][a] < 0
                assert abs(sc) == 2.0 ** round(np.log2(abs(sc)))
                assert np.array_equal(table[:, k], sc * table[:, a]), (name, per_node, k, a, sc)
    assert found["M3_film"] == 3 and found["M5_stiff"] == 3 and found["burgers"] == 0, found


def bound_solver(m, fd, pars, **opts):
    cm = m._device
    N = fd["x"].size
    solver = cm.solver(N, pars["periodic"], 1, 0, **opts)
    cm.bind_inputs(solver, fd["x"], [pars[k] for k in cm.pars],
                   [fd[k] for k in m._help_funcs] if cm.nh else None)
    solver.set_state(0, np.array([fd[k] for k in m._dep_vars]))
    return solver


def check_linear_solve(name, backend, N, level_opts, c=0.01, tol=1e-9):
    """(I - cJ) x = b against SuperLU (the reference's solver, schemes.py:149,557)
    for every level plan in ``level_opts``, periodic and clamped."""
    m, mo = device_model(name, backend), oracle_model(name)
    rng = np.random.default_rng(5)
    for periodic in (True, False):
        fd = corpus.synthetic_fields(name, N, seed=7, periodic=periodic, length=N * 5e-3)
        pars = corpus.synthetic_pars(name, N, periodic)
        Jo = mo.J(mo.fields_template(**fd), pars)
        n = N * m._nvar
        A = sps.identity(n, format="csc") - c * Jo
        rhs = rng.standard_normal(n)
        xs = spla.spsolve(A, rhs)
        for opts in level_opts:
            solver = bound_solver(m, fd, pars, **opts)
            solver.eval(0, with_j=True)
            solver.factor(c)
            x = solver.solve(rhs)[0]
            err = np.abs(x - xs).max() / np.abs(xs).max()
            assert err <= tol, (name, periodic, opts, solver.describe(), err)
            # a second factorisation of the same matrix starts from the pivot orders the first one
            # stored (cyclic-reduction levels, tf_gj_node): the other code path, the same answer
            solver.factor(c)
            x2 = solver.solve(rhs)[0]
            err2 = np.abs(x2 - xs).max() / np.abs(xs).max()
            assert err2 <= tol, (name, periodic, opts, solver.describe(), "second factorisation", err2)
            y = solver.matvec(rhs)[0]
            assert np.abs(y - Jo @ rhs).max() <= 1e-12 * max(1.0, np.abs(Jo @ rhs).max())


def check_respike(backend):
    """Level-1 solves without the stored spike response (a.respike: second elimination of the
    right-hand side with the separator above known, back-substitution with U alone -- the form
    large multi-variable problems use) against the stored-E form and against SuperLU: plain solves
    on several level plans, periodic and clamped, and steps whose first solve rides with the
    factorisation, with a Dirichlet hook and with two members."""
    import os
    from triflow_amd.ensemble import Ensemble
    rng = np.random.default_rng(11)

    def with_env(value, fn):
        os.environ["TRIFLOW_L1_RESPIKE"] = value
        try:
            return fn()
        finally:
            del os.environ["TRIFLOW_L1_RESPIKE"]

    for name, N in (("M3_film", 1203), ("M5_stiff", 811), ("pair4", 403), ("six", 333)):
        m, mo = device_model(name, backend), oracle_model(name)
        for periodic in (True, False):
            fd = corpus.synthetic_fields(name, N, seed=3, periodic=periodic, length=N * 5e-3)
            pars = corpus.synthetic_pars(name, N, periodic)
            Jo = mo.J(mo.fields_template(**fd), pars)
            n = N * m._nvar
            c = 0.01
            rhs = rng.standard_normal(n)
            xs = spla.spsolve(sps.identity(n, format="csc") - c * Jo, rhs)
            # chunk lengths on both sides of the twisted form's limits (tf_twist_h: 4 mp interior nodes)
            for opts in (dict(refine=0), dict(refine=0, m1=8, m_upper=4), dict(refine=0, m1=5),
                         dict(refine=0, m1=32), dict(refine=0, m1=13, m_upper=5), dict(refine=0, m1=10),
                         # (chunks of m1 and m1 + 1 nodes in one level: twisted next to one-sided ones)
                         dict(refine=0, m1=9), dict(refine=0, m1=4)):
                xv = []
                for flag in ("1", "0"):
                    def run():
                        solver = bound_solver(m, fd, pars, **opts)
                        solver.eval(0, with_j=True)
                        solver.factor(c)
                        return solver.solve(rhs)[0]
                    xv.append(with_env(flag, run))
                e1 = np.abs(xv[0] - xs).max() / np.abs(xs).max()
                e0 = np.abs(xv[1] - xs).max() / np.abs(xs).max()
                # both are backward stable; neither may be much further from SuperLU than the other
                assert e1 <= max(1e-9, 50 * e0), (name, periodic, opts, e1, e0)
    for cfg, sch, hook, N, nsys in ((3, "ROS2", None, 3001, 2), (3, "RODASPR", None, 1501, 1),
                                    (5, "BDF2", DEVICE_HOOKS["cfg5"], 2003, 1)):
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: np.repeat(v[None, :], nsys, axis=0) * (1 + 0.01 * np.arange(nsys))[:, None]
                  for k, v in fd.items() if k != "x"}
        out = []
        for flag in ("1", "0"):
            def run():
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2,
                               m1=32)
                for _ in range(6):
                    ens.step(dt)
                ens.sync()
                st = ens.state().copy()
                ens.close()
                return st
            out.append(with_env(flag, run))
        err = np.abs(out[0] - out[1]).max() / np.abs(out[1]).max()
        print("respike vs stored E, config %d %s: %.1e" % (cfg, sch, err))
        assert np.isfinite(out[0]).all() and err <= 1e-10, (cfg, sch, err)


def check_bdf2_history_in_place(backend):
    """BDF-2 through an Ensemble with three rotating state slots or more reads U_{n-1} in the slot
    the previous step started from (tf_step_bdf2_from: no copy of the history); with two slots the
    solver keeps a copy (tf_step_bdf2).  The same states, bit for bit: fixed steps, a change of
    the step size (the history is dropped: backward-Euler form), a restart, with and without hook."""
    from triflow_amd.ensemble import Ensemble
    for hook, N, nsys in ((DEVICE_HOOKS["cfg5"], 1203, 1), (None, 611, 2)):
        name, fd, pars, dt, _ = corpus.config_inputs(5, N)
        m = device_model(name, backend)
        fields = {k: np.repeat(v[None, :], nsys, axis=0) * (1 + 0.01 * np.arange(nsys))[:, None]
                  for k, v in fd.items() if k != "x"}
        out = []
        for nstate in (5, 4, 3):             # rotation over 4 and 3 slots (one keeps the start), over 2
            ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme="BDF2", hook=hook, nstate=nstate)
            assert (ens._nrot >= 3) == (nstate >= 4)
            states = []
            for k in range(9):
                ens.step(dt if k < 5 else 0.5 * dt)
                if k in (0, 1, 4, 5, 8):
                    ens.sync()
                    states.append(ens.state().copy())
            ens.restart()
            for _ in range(3):
                ens.step(dt)
            ens.sync()
            states.append(ens.state().copy())
            ens.close()
            out.append(states)
        for other in out[1:]:
            for a, b in zip(out[0], other):
                assert np.isfinite(a).all() and np.array_equal(a, b)


def check_bdf2_history_is_the_hooked_state(backend):
    """The history of a BDF-2 step is the *hooked* input of the step before (the oracle keeps the
    hooked copy, oracle/numpy_path.py BDF2._prev).  An initial -- or uploaded -- state whose boundary
    nodes differ from the hook's values: the rotation over state slots (tf_step_bdf2_from) against the
    oracle scheme and against the copied history, including a state uploaded between two steps."""
    from triflow_amd.ensemble import Ensemble
    N = 301
    name, fd, pars, dt, _ = corpus.config_inputs(5, N)
    hook_values = {0: 2.0, -1: 2.0}                       # the initial A is 1 everywhere
    hook = DirichletHook(A=dict(hook_values))
    def ora_hook(t, fields, pars):
        for node, value in hook_values.items():
            fields["A"][node] = value
        return fields, pars
    m, mo = device_model(name, backend), oracle_model(name)
    sch = ora.BDF2(mo)
    fo = mo.fields_template(**fd)
    ref = []
    t = 0.0
    for k in range(4):
        t, fo = sch(t, fo, dt, pars, hook=ora_hook)
        ref.append(np.array([np.asarray(fo[v]) for v in mo._dep_vars]))
    fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
    for nstate in (4, 3):
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme="BDF2", hook=hook, nstate=nstate)
        for k in range(4):
            ens.step(dt)
            ens.sync()
            got = ens.state()[:, 0, :]
            err = np.abs(got - ref[k]).max() / np.abs(ref[k]).max()
            assert err <= 1e-9, (nstate, k, err)
        ens.close()
    # a state uploaded between two steps (its boundary nodes violate the hook again): the step after
    # the upload still uses the hooked upload as U_n and the hooked U_{n-1} as history
    outs = []
    for nstate in (4, 3):
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme="BDF2", hook=hook, nstate=nstate)
        ens.step(dt); ens.step(dt); ens.sync()
        up = ens.state().copy()
        up[0, :, 0] = 1.5
        up[0, :, -1] = 0.5
        ens.solver.set_state(ens.cur, up)
        ens.step(dt); ens.step(dt); ens.sync()
        outs.append(ens.state().copy())
        ens.close()
    assert np.isfinite(outs[0]).all()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-12 * np.abs(outs[1]).max()


def check_hook_input_in_place(backend):
    """A step with a Dirichlet hook starts from a copy of the state with the hook applied
    (schemes.py:144-145, 548-549).  The device skips the copy when the source slot is what an
    earlier step left with the hook applied at the same time and nothing has written it since
    (tf_solver::slot_hook): the same states as with the copy (TRIFLOW_HOOK_IN_PLACE=0) -- constant and
    time-dependent boundary values, a state uploaded in between (the next step copies again), a
    restart, and the scheme objects driven from Python."""
    import os
    from triflow_amd.ensemble import Ensemble

    def with_env(value, fn):
        os.environ["TRIFLOW_HOOK_IN_PLACE"] = value
        try:
            return fn()
        finally:
            del os.environ["TRIFLOW_HOOK_IN_PLACE"]

    moving = DirichletHook(A={0: lambda t: 1.0 + 0.5 * t, -1: 1.0})
    for cfg, sch, hook, N in ((5, "BDF2", DEVICE_HOOKS["cfg5"], 1203), (1, "Theta", DEVICE_HOOKS["cfg1"], 200),
                              (5, "ROS2", DEVICE_HOOKS["cfg5"], 803), (5, "BDF2", moving, 611)):
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        out = []
        for flag in ("1", "0"):
            def run():
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=3)
                states = []
                for _ in range(4):
                    ens.step(dt)
                ens.sync()
                states.append(ens.state().copy())
                # a state from outside: the slot no longer holds what a step left
                ens.solver.set_state(ens.cur, states[0] * 1.001)
                for _ in range(3):
                    ens.step(dt)
                ens.sync()
                states.append(ens.state().copy())
                ens.restart()
                for _ in range(3):
                    ens.step(dt)
                ens.sync()
                states.append(ens.state().copy())
                ens.close()
                return states
            out.append(with_env(flag, run))
        for a, b in zip(*out):
            assert np.isfinite(a).all() and np.array_equal(a, b), (cfg, sch)
    # the scheme objects (fields in, fields out): five steps with the hook, against the copy form
    for flag_states in [[]]:
        name, fd, pars, dt, _ = corpus.config_inputs(5, 403)
        for flag in ("1", "0"):
            def run():
                m = device_model(name, backend)
                scheme = schemes.BDF2(m)
                fields = m.fields_template(**fd)
                t = 0.0
                for _ in range(5):
                    t, fields = scheme(t, fields, dt, pars, hook=DEVICE_HOOKS["cfg5"])
                return np.array(fields.uflat)
            flag_states.append(with_env(flag, run))
        assert np.array_equal(flag_states[0], flag_states[1])


# ---------------------------------------------------------------- seam #2: schemes
DEVICE_SCHEMES = {
    "Theta1": lambda m: schemes.Theta(m, theta=1),
    "Theta05": lambda m: schemes.Theta(m, theta=0.5),
    "Theta0": lambda m: schemes.Theta(m, theta=0),
    "ROS2": lambda m: schemes.ROS2(m),
    "ROS3PRw": lambda m: schemes.ROS3PRw(m, time_stepping=False),
    "ROS3PRL": lambda m: schemes.ROS3PRL(m, time_stepping=False),
    "RODASPR": lambda m: schemes.RODASPR(m, time_stepping=False),
    "ROS3PRw_adapt": lambda m: schemes.ROS3PRw(m, tol=1e-1),
    "ROS3PRL_adapt": lambda m: schemes.ROS3PRL(m, tol=1e-1),
    "RODASPR_adapt": lambda m: schemes.RODASPR(m, tol=1e-1),
}
#: per-case relative tolerance on U after each of the five steps.  The solver is
#: not SuperLU, so agreement is bounded by cond(A) * eps of BOTH solvers: the
#: film model at the benchmark dx has cond ~ 1e10 (DESIGN.md "tolerances").
STEP_TOL = {"cfg1": 1e-11, "cfg1_nohook": 1e-11, "diff_per": 1e-11, "burgers_per": 1e-11,
            "film_per": 2e-7, "film_clamp": 2e-7, "stiff_clamp": 1e-9}
DEVICE_HOOKS = {"cfg1": DirichletHook(U={0: 1.0, -1: 0.0}),
                "cfg5": DirichletHook(A={0: 1.0, -1: 1.0}), None: None}


def check_steps_golden(case, backend, python_hook=False, only=None):
    """G2: five steps of every scheme against the reference's trajectory."""
    g = np.load(os.path.join(GOLDEN, "steps.npz"))
    cname, mname, N, periodic, dt, hook = case
    m = device_model(mname, backend)
    fdict, pars = step_inputs(case)
    tol = STEP_TOL[cname]
    for sname, make in DEVICE_SCHEMES.items():
        key = "%s|%s" % (cname, sname)
        if key not in g.files or (only and sname not in only):
            continue
        scheme = make(m)
        fields = m.fields_template(**fdict)
        t = 0.0
        kw = {}
        if hook:
            kw["hook"] = HOOKS[hook] if python_hook else DEVICE_HOOKS[hook]
        with np.errstate(all="ignore"):
            for k in range(5):
                t, fields = scheme(t, fields, dt, pars, **kw)
                ref = g[key][k]
                err = np.abs(fields.uflat - ref).max() / np.abs(ref).max()
                assert err <= tol, (key, k, err)
        assert np.isclose(t, 5 * dt)


def check_adaptive_landing_reuse(backend):
    """The default user path: adaptive Rosenbrock steps driven to a target (schemes.py:176-238).  Once the
    controller's own step exceeds the caller's dt, the reference's landing step repeats the accepted
    trial with a dt that differs by the rounding of t + dt; the device scheme takes the trial's state
    instead (ROW_general.REUSE_TRIAL_AS_LANDING): same states to 1e-13 as the literal sequence, half the
    Rosenbrock steps -- no hook, the declarative hook, a Python hook (never reused).  And a
    constant-matrix Theta scheme driven by Simulation (dt = target - t, an ulp off the last one) keeps its
    factorisation."""
    from triflow_amd import _capi
    calls = {"row": 0}
    orig, orig_q = _capi.DeviceSolver.step_row, _capi.DeviceSolver.step_row_queued

    def counting(self, *a, **k):
        calls["row"] += 1
        return orig(self, *a, **k)

    def counting_q(self, *a, **k):
        calls["row"] += 1
        return orig_q(self, *a, **k)
    _capi.DeviceSolver.step_row, _capi.DeviceSolver.step_row_queued = counting, counting_q
    try:
        for cfg, N, hook in ((3, 400, None), (1, 200, DEVICE_HOOKS["cfg1"]), (1, 200, HOOKS["cfg1"])):
            name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
            m = device_model(name, backend)
            out, counts = [], []
            for reuse in (True, False):
                scheme = schemes.RODASPR(m, tol=1e-1)
                scheme.REUSE_TRIAL_AS_LANDING = reuse
                f, t = m.fields_template(**fd), 0.3
                kw = {"hook": hook} if hook is not None else {}
                calls["row"] = 0
                states = []
                with np.errstate(all="ignore"):
                    for k in range(14):
                        t, f = scheme(t, f, dt, pars, **kw)
                        states.append(f.uflat.copy())
                out.append(states)
                counts.append(calls["row"])
            for a, b in zip(*out):
                assert np.isfinite(a).all()
                assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max(), (cfg, np.abs(a - b).max())
            if hook is HOOKS["cfg1"]:
                assert counts[0] == counts[1]                  # a Python hook: the literal sequence
            else:
                assert counts[0] < counts[1], counts           # fewer Rosenbrock steps
                assert counts[1] - counts[0] >= 3, counts      # ... by one per call once the controller's step exceeds dt
        # The first trial of the next call queued ahead of the read of this call's error estimate
        # (ROW_general.QUEUE_NEXT_TRIAL): the same steps, earlier -- bit for bit the states of the
        # controller that launches every trial when it needs it, also when the caller changes dt, hands
        # in a copy of the container, or other parameters (the queued step is dropped)
        name, fd, pars, dt, _ = corpus.config_inputs(3, 400)
        m = device_model(name, backend)
        out = []
        for queue in (True, False):
            scheme = schemes.RODASPR(m, tol=1e-1)
            scheme.QUEUE_NEXT_TRIAL = queue
            f, t, p = m.fields_template(**fd), 0.3, pars
            states = []
            with np.errstate(all="ignore"):
                for k in range(22):
                    if k == 14:
                        f = f.copy()                     # another container: nothing queued applies
                    if k == 17:
                        p = dict(p, We=0.02)             # other parameters
                    t, f = scheme(t, f, dt if k < 10 or k >= 12 else 0.5 * dt, p)
                    states.append(f.uflat.copy())
            out.append(states)
        for a, b in zip(*out):
            assert np.isfinite(a).all() and np.array_equal(a, b)
    finally:
        _capi.DeviceSolver.step_row, _capi.DeviceSolver.step_row_queued = orig, orig_q
    # Simulation -> Theta on the constant-matrix model: one factorisation serves every landing step
    name, fd, pars, dt, _ = corpus.config_inputs(2, 2000)
    m = device_model(name, backend)
    sim = Simulation(m, dict(fd), pars, dt, scheme=schemes.Theta, time_stepping=False)
    it = iter(sim)
    for _ in range(3):
        next(it)
    solver = sim.fields._device_backing().stepper.solver
    c0 = solver.counters()["factorisations"]
    for _ in range(12):
        next(it)
    assert solver.counters()["factorisations"] == c0


def check_bdf2(backend):
    """BDF-2 (not in the reference): device scheme against the oracle's
    restatement, same F/J, SuperLU vs banded solver."""
    for mname, N, periodic, dt, hook in (("M2_diff", 40, True, 1e-2, None),
                                         ("M5_stiff", 24, False, 1e-3, "cfg5")):
        m, mo = device_model(mname, backend), oracle_model(mname)
        if mname == "M5_stiff":
            _, fd, pars, _, _ = corpus.config_inputs(5, N)
        else:
            fd = corpus.synthetic_fields(mname, N, seed=2, periodic=periodic)
            pars = corpus.synthetic_pars(mname, N, periodic)
        s_dev, s_ora = schemes.BDF2(m), ora.BDF2(mo)
        f_dev, f_ora = m.fields_template(**fd), mo.fields_template(**fd)
        t = 0.0
        for k in range(6):
            kw_d = dict(hook=DEVICE_HOOKS[hook]) if hook else {}
            kw_o = dict(hook=HOOKS[hook]) if hook else {}
            _, f_dev = s_dev(t, f_dev, dt, pars, **kw_d)
            t, f_ora = s_ora(t, f_ora, dt, pars, **kw_o)
            err = np.abs(f_dev.uflat - f_ora.uflat).max() / np.abs(f_ora.uflat).max()
            assert err <= 1e-10, (mname, k, err)


def check_bdf2_against_vode(backend):
    """The DEVICE BDF-2 against the reference-generated anchor (tests/golden/vode_bdf.npz: the
    reference's scipy_ode(vode, bdf) trajectory of the heat equation, oracle/gen_golden.py): close
    to it at the finest step and converging with order 2 -- the check test_oracle_golden.py makes
    for the oracle's restatement, here on the scheme the product ships."""
    g = np.load(os.path.join(GOLDEN, "vode_bdf.npz"))
    m = device_model("M2_diff", backend)
    x = g["x"]
    pars = dict(periodic=True, k=1)
    errs = []
    for nsub in (10, 20, 40):
        scheme = schemes.BDF2(m)
        fields = m.fields_template(x=x, U=np.cos(x * 2 * np.pi / 10))
        t = 0.0
        for _ in range(10 * nsub):
            t, fields = scheme(t, fields, 0.1 / nsub, pars)
        errs.append(np.abs(fields.uflat - g["U"][-1]).max())
    assert errs[0] < 1e-3
    order = np.log2(errs[0] / errs[1]), np.log2(errs[1] / errs[2])
    assert 1.6 < order[0] < 2.4 and 1.6 < order[1] < 2.4, (errs, order)


def check_bdf2_interleaved(backend):
    """Two BDF2 objects stepping alternately on one model share one device solver; each
    keeps its own history U_{n-1}: both trajectories equal their solo runs bit for bit.  A
    scheme object restarted from other fields starts with the backward-Euler form again."""
    mname, N, dt = "M2_diff", 64, 1e-2
    m = device_model(mname, backend)
    pars = corpus.synthetic_pars(mname, N, True)
    fds = [corpus.synthetic_fields(mname, N, seed=sd, periodic=True) for sd in (2, 5)]

    def solo(fd, nsteps):
        sch, f, t = schemes.BDF2(m), m.fields_template(**fd), 0.0
        for _ in range(nsteps):
            t, f = sch(t, f, dt, pars)
        return f.uflat.copy()

    want = [solo(fd, 5) for fd in fds]
    a, b = schemes.BDF2(m), schemes.BDF2(m)
    fa, fb, t = m.fields_template(**fds[0]), m.fields_template(**fds[1]), 0.0
    for _ in range(5):
        _, fa = a(t, fa, dt, pars)
        t, fb = b(t, fb, dt, pars)
    assert np.array_equal(fa.uflat, want[0]) and np.array_equal(fb.uflat, want[1])
    assert not np.array_equal(want[0], want[1])
    # restart of a used scheme object from new fields == a fresh object
    f, t = m.fields_template(**fds[1]), 0.0
    for _ in range(5):
        t, f = a(t, f, dt, pars)
    assert np.array_equal(f.uflat, want[1])
    # the two-step formula is really in use (differs from a chain of backward-Euler steps)
    f, t = m.fields_template(**fds[0]), 0.0
    for _ in range(5):
        t, f = schemes.BDF2(m)(t, f, dt, pars)
    assert not np.array_equal(f.uflat, want[0])


def check_simulation_golden(backend):
    """G3: Simulation on config 1, device Theta scheme, python and declarative hook,
    with and without the default step-doubling wrapper."""
    g = np.load(os.path.join(GOLDEN, "simulation.npz"))
    for ts in (False, True):
        for hook in (corpus.dirichlet_hook_cfg1, DEVICE_HOOKS["cfg1"]):
            m = device_model("M1_advdiff", backend)
            _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
            sim = Simulation(m, fdict, pars, dt, hook=hook, tmax=2.5, scheme=schemes.Theta,
                             time_stepping=ts)
            us, tl = [], []
            for t, fields in sim:
                tl.append(t)
                us.append(fields.uflat.copy())
            assert np.allclose(tl, g["Theta_ts%i_t" % ts], rtol=0, atol=1e-12)
            assert np.abs(np.array(us) - g["Theta_ts%i_U" % ts]).max() <= 1e-10, (ts, hook)


def check_resident_fields(backend):
    """Device-backed containers: no upload when handed back untouched, snapshots
    stay valid while newer steps recycle the slots, inputs are never mutated."""
    m = device_model("M2_diff", backend)
    fd = corpus.synthetic_fields("M2_diff", 64, seed=1)
    pars = corpus.synthetic_pars("M2_diff", 64, True)
    scheme = schemes.Theta(m)
    f0 = m.fields_template(**fd)
    u0 = f0.uflat.copy()
    t, f1 = scheme(0.0, f0, 1e-2, pars)
    assert f1._device_backing() is not None
    assert np.array_equal(f0.uflat, u0)
    snaps = [f1]
    f = f1
    for _ in range(7):
        t, f = scheme(t, f, 1e-2, pars)
        snaps.append(f)
    stepper = f._device_backing().stepper
    assert stepper.nstate < len(snaps)
    # replay on the host copies: identical trajectories
    g = m.fields_template(**fd)
    for k in range(8):
        _, g = scheme(0.0, g, 1e-2, pars)
        g["U"]                                   # force the host copy each step
        assert np.array_equal(np.asarray(snaps[k]["U"]), np.asarray(g["U"])), k
    c = snaps[-1].copy()
    assert np.array_equal(np.asarray(c["U"]), np.asarray(snaps[-1]["U"]))


def check_errors(backend):
    """Error behaviour of the reference seams: KeyError for a missing parameter
    (routines.py:11,40), RuntimeError from the step control (schemes.py:229-238)."""
    import pytest
    m = device_model("M2_diff", backend)
    x = np.linspace(0, 10, 50, endpoint=False)
    fields = m.fields_template(x=x, U=np.cos(x * 2 * np.pi / 10))
    with pytest.raises(KeyError):
        m.F(fields, dict(periodic=True))
    with pytest.raises(KeyError):
        schemes.Theta(m)(0, fields, 1.0, dict(k=1.0))
    pars = dict(periodic=True, k=1)
    sim = Simulation(m, fields, pars, dt=1, tol=1e-1, max_iter=2)
    with pytest.raises(RuntimeError):
        for _ in sim:
            pass
    assert sim.status == "failed"
    sim = Simulation(m, fields, pars, dt=1, tol=1e-1, dt_min=.1)
    with pytest.raises(RuntimeError):
        for _ in sim:
            pass


def check_heat_steady_state(backend, scheme_cls, dirichlet):
    """The reference's own integration tests (tests/test_simulation.py:20-58)."""
    m = device_model("M2_diff", backend)
    x = np.linspace(0, 10, 50, endpoint=False)
    fields = m.fields_template(x=x, U=np.cos(x * 2 * np.pi / 10))
    if dirichlet:
        def hook(t, fields, parameters):
            fields["U"][0] = 1
            fields["U"][-1] = 1
            return fields, parameters
        sim = Simulation(m, fields, dict(periodic=False, k=1), hook=hook, scheme=scheme_cls,
                         dt=.1 if not isinstance(dirichlet, float) else dirichlet,
                         tmax=100 if dirichlet is True else 20, tol=1e-1)
        for t, f in sim:
            pass
        assert np.isclose(t, sim.tmax)
        if dirichlet is True:
            assert np.isclose(np.asarray(f["U"]), 1, atol=1e-1).all()
    else:
        sim = Simulation(m, fields, dict(periodic=True, k=1), scheme=scheme_cls, dt=1,
                         tmax=100, tol=1e-1)
        for t, f in sim:
            pass
        assert t == 100
        assert np.isclose(np.asarray(f["U"]).mean(), 0)


def check_step_doubling_device_norm(backend):
    """The step-doubling wrapper with the error norm reduced on the device
    (containers untouched on the host) against the same run with host norms."""
    m = device_model("M2_diff", backend)
    fd = corpus.synthetic_fields("M2_diff", 80, seed=4)
    pars = corpus.synthetic_pars("M2_diff", 80, True)
    out = []
    for force_host in (False, True):
        scheme = schemes.time_stepping(schemes.Theta(m), tol=1e-2)
        f, t = m.fields_template(**fd), 0.0
        for _ in range(3):
            t, f = scheme(t, f, 0.05, pars)
            if force_host:
                f["U"]                     # materialise: next call uploads, norms on host
        out.append(f.uflat.copy())
    assert np.abs(out[0] - out[1]).max() <= 1e-12 * np.abs(out[1]).max()
    # the device-backed run really kept the norm on the GPU
    a, b = m.fields_template(**fd), m.fields_template(**fd)
    _, fa = schemes.Theta(m)(0.0, a, 0.01, pars)
    _, fb = schemes.Theta(m)(0.0, fa, 0.01, pars)
    n_dev = schemes._difference_norms(fa, fb, 2)
    assert fa._device_backing() is not None and fb._device_backing() is not None
    n_host = [np.linalg.norm(np.asarray(fa["U"]) - np.asarray(fb["U"]), 2)]
    assert np.allclose(n_dev, n_host, rtol=1e-13)
    n_dev_inf = None
    _, fc = schemes.Theta(m)(0.0, fb, 0.01, pars)
    n_dev_inf = schemes._difference_norms(fb, fc, np.inf)
    assert np.allclose(n_dev_inf, [np.abs(np.asarray(fb["U"]) - np.asarray(fc["U"])).max()], rtol=0)


def check_fused_step_doubling(backend):
    """Row f1: a trial of the step-doubling controller as one device call
    (tf_step_doubling: coarse step + ten fine steps + norm, one host wait) gives the bits of
    the same trial driven step by step from Python, for Theta and a fixed-step Rosenbrock
    scheme, with and without a constant Dirichlet hook; and it really is one call."""
    from triflow_amd import _capi
    cases = [("M2_diff", 80, True, lambda m: schemes.Theta(m), None, 0.05),
             ("M1_advdiff", 64, False, lambda m: schemes.Theta(m, theta=0.5), "cfg1", 0.5),
             ("M3_film", 48, True, lambda m: schemes.ROS2(m), None, 1e-3)]
    for mname, N, periodic, make, hook, dt in cases:
        m = device_model(mname, backend)
        fd = corpus.synthetic_fields(mname, N, seed=4, periodic=periodic)
        pars = corpus.synthetic_pars(mname, N, periodic)
        kw = dict(hook=DEVICE_HOOKS[hook]) if hook else {}
        out, calls = [], []
        for fused in (True, False):
            counts = dict(single=0, doubling=0)
            orig = {k: getattr(_capi.DeviceSolver, k) for k in ("step_theta", "step_row", "step_doubling")}

            def counted(name, key):
                def f(self, *a, **k):
                    counts[key] += 1
                    return orig[name](self, *a, **k)
                return f
            _capi.DeviceSolver.step_theta = counted("step_theta", "single")
            _capi.DeviceSolver.step_row = counted("step_row", "single")
            _capi.DeviceSolver.step_doubling = counted("step_doubling", "doubling")
            saved, saved_flag = schemes._fused_trial, schemes.FUSED_TRIALS
            schemes.FUSED_TRIALS = True               # (off by default: the fused call is not faster)
            if not fused:
                schemes._fused_trial = lambda *a, **k: None
            try:
                scheme = schemes.time_stepping(make(m), tol=1e-2)
                f, t = m.fields_template(**fd), 0.0
                for _ in range(3):
                    t, f = scheme(t, f, dt, pars, **kw)
                out.append((t, f.uflat.copy()))
            finally:
                schemes._fused_trial, schemes.FUSED_TRIALS = saved, saved_flag
                for k, v in orig.items():
                    setattr(_capi.DeviceSolver, k, v)
            calls.append(counts)
        assert out[0][0] == out[1][0], (mname, out[0][0], out[1][0])
        assert np.array_equal(out[0][1], out[1][1]), mname
        assert calls[0]["doubling"] >= 1 and calls[1]["doubling"] == 0, calls
        assert calls[1]["single"] >= calls[0]["single"] + 11 * calls[0]["doubling"], calls


def check_time_dependent_hook(backend):
    """Time-dependent Dirichlet data and a time-dependent parameter, served on
    the device by a declarative hook, against the oracle running the same hook as
    a plain Python callable on the host."""
    values = dict(U={0: lambda t: 1.0 + 0.5 * np.sin(3 * t), -1: 0.25})
    kfun = lambda t, pars: dict(k=1e-3 * (1.0 + t))
    dev_hook = DirichletHook(parameters=kfun, **values)

    def host_hook(t, fields, pars):
        fields["U"][0] = 1.0 + 0.5 * np.sin(3 * t)
        fields["U"][-1] = 0.25
        return fields, dict(pars, **kfun(t, pars))

    for make_dev, make_ora in ((lambda m: schemes.Theta(m), lambda m: ora.Theta(m)),
                               (lambda m: schemes.ROS2(m), lambda m: ora.ROS2(m)),
                               (lambda m: schemes.ROS3PRL(m, time_stepping=False),
                                lambda m: ora.ROS3PRL(m, time_stepping=False))):
        m, mo = device_model("M1_advdiff", backend), oracle_model("M1_advdiff")
        _, fd, pars, dt, _ = corpus.config_inputs(1, 60)
        s_dev, s_ora = make_dev(m), make_ora(mo)
        f_dev, f_ora = m.fields_template(**fd), mo.fields_template(**fd)
        t = 0.0
        for k in range(4):
            _, f_dev = s_dev(t, f_dev, dt, pars, hook=dev_hook)
            assert f_dev._device_backing() is not None        # never left the GPU
            t, f_ora = s_ora(t, f_ora, dt, pars, hook=host_hook)
        err = np.abs(f_dev.uflat - f_ora.uflat).max() / np.abs(f_ora.uflat).max()
        assert err <= 1e-11, err


#: the models of the reference's example notebooks (examples/notebooks/*.ipynb) with
#: their grids, parameters and time steps
NOTEBOOK_CASES = {
    "burgers_kdv": (("-U * dxU + a * dxxU + b * dxxxU", "U", ["a", "b"]),
                    lambda: np.linspace(-2, 6, 5000, endpoint=False),
                    dict(a=2e-4, b=1e-4, periodic=False), 0.05),
    "kuramoto": (("-dxxzeta - dxxxxzeta + (dxzeta)**2", "zeta"),
                 lambda: np.linspace(0, 200, 2010), dict(periodic=True), 0.2),
    "droplet": (("dx((h**3 + h**2) * dx(-sigma * dxxh + alpha * (1 / h**3 - e / h**4)))",
                 "h", ["sigma", "alpha", "e"]),
                lambda: np.linspace(0, 10, 200),
                dict(periodic=False, alpha=.05, sigma=10, e=1e-1), 0.01),
    "so_wavy": ((["k * dxxU - c * U * dxV", "k * dxxV - c * V * dxU"], ["U", "V"], ["k", "c"]),
                lambda: np.linspace(0, 100, 500, endpoint=False),
                dict(k=1, c=10, periodic=True), 0.1),
}


def check_notebook_model(name, backend):
    """Three steps of Theta and of fixed-step RODASPR on a model of the reference's
    example notebooks, device path vs oracle."""
    margs, grid, pars, dt = NOTEBOOK_CASES[name]
    x = grid()
    compiler = hip_compiler if backend is None else partial(hip_compiler, backend=backend)
    m = Model(*margs, compiler=compiler)
    mo = Model(*margs, compiler=ora.numpy_compiler)
    if name == "burgers_kdv":
        fd = dict(x=x, U=np.exp(-x ** 2 * 4))
    elif name == "kuramoto":
        fd = dict(x=x, zeta=np.cos(x * 2 * np.pi / x.max() * 10) * 2 + 5)
    elif name == "droplet":
        fd = dict(x=x, h=np.exp(-0.5 * ((x - 5) / 1.0) ** 2) + 1e-1)
    else:
        fd = dict(x=x, U=np.cos(x * 2 * np.pi / 100 * 3), V=np.sin(x * 2 * np.pi / 100 * 2))
    F = m.F(m.fields_template(**fd), pars)
    Fo = mo.F(mo.fields_template(**fd), pars)
    if name == "droplet":
        # h**3, h**4, h**5 of a *field*: NumPy calls its libm / SVML pow (not correctly
        # rounded, and not the same on every CPU); the kernel rounds the exact power
        # once.  1-ulp differences of those terms are amplified by cancellation.
        # Bound: 4 ulp of the largest term, 6*sigma*h**4/dx**4.
        dx = (x[-1] - x[0]) / (x.size - 1)
        big = 6 * pars["sigma"] * fd["h"].max() ** 4 / dx ** 4
        assert np.abs(F - Fo).max() <= 4 * np.spacing(big)
    else:
        assert np.array_equal(F, Fo)
    for mk_d, mk_o in ((lambda mm: schemes.Theta(mm), lambda mm: ora.Theta(mm)),
                       (lambda mm: schemes.RODASPR(mm, time_stepping=False),
                        lambda mm: ora.RODASPR(mm, time_stepping=False))):
        sd, so = mk_d(m), mk_o(mo)
        f_d, f_o = m.fields_template(**fd), mo.fields_template(**fd)
        t = 0.0
        for _ in range(3):
            _, f_d = sd(t, f_d, dt, pars)
            t, f_o = so(t, f_o, dt, pars)
        err = np.abs(f_d.uflat - f_o.uflat).max() / np.abs(f_o.uflat).max()
        assert err <= 1e-9, (name, err)


def check_simulation_stays_resident(backend):
    """README-style run with a declarative hook: the state is uploaded once and only
    downloaded when the user reads it (here: once, at the end)."""
    from triflow_amd import _capi
    m = device_model("M1_advdiff", backend)
    _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
    counts = dict(up=0, down=0)
    orig_set, orig_get = _capi.DeviceSolver.set_state, _capi.DeviceSolver.get_state

    def set_state(self, *a, **k):
        counts["up"] += 1
        return orig_set(self, *a, **k)

    def get_state(self, *a, **k):
        counts["down"] += 1
        return orig_get(self, *a, **k)
    _capi.DeviceSolver.set_state, _capi.DeviceSolver.get_state = set_state, get_state
    try:
        sim = Simulation(m, fdict, pars, dt, hook=DEVICE_HOOKS["cfg1"], tmax=2.5,
                         scheme=schemes.Theta, time_stepping=False)
        for t, fields in sim:
            pass
        assert counts == dict(up=1, down=0), counts
        U = fields.uflat
        assert counts == dict(up=1, down=1), counts
        for ts in (True,):
            sim = Simulation(m, fdict, pars, dt, hook=DEVICE_HOOKS["cfg1"], tmax=2.5,
                             scheme=schemes.ROS2, time_stepping=ts)
            before = dict(counts)
            for t, fields in sim:
                pass
            assert counts["down"] == before["down"], counts      # norms reduced on the device
    finally:
        _capi.DeviceSolver.set_state, _capi.DeviceSolver.get_state = orig_set, orig_get
    g = np.load(os.path.join(GOLDEN, "simulation.npz"))
    assert np.abs(U - g["Theta_ts0_U"][-1]).max() <= 1e-10


def check_container_on_device_fields(backend, tmp_dir):
    """Row f3: a persistence container attached to a run whose fields live on the device.
    Every emitted state is saved (one download per snapshot, nothing else), the files are
    the reference's layout, the saved trajectory equals the reference golden sequence, and
    the run itself is not disturbed (same final state as without a container)."""
    from triflow_amd import _capi, retrieve_container
    m = device_model("M1_advdiff", backend)
    _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
    downs = []
    orig_get = _capi.DeviceSolver.get_state

    def get_state(self, *a, **k):
        downs.append(1)
        return orig_get(self, *a, **k)
    _capi.DeviceSolver.get_state = get_state
    try:
        sim = Simulation(m, fdict, pars, dt, hook=DEVICE_HOOKS["cfg1"], tmax=2.5,
                         scheme=schemes.Theta, time_stepping=False, id="dev_run")
        c = sim.attach_container(str(tmp_dir), nbuffer=2)
        sim.run(progress=False)
        assert len(downs) == 5, len(downs)              # 5 device-resident snapshots, 5 downloads
    finally:
        _capi.DeviceSolver.get_state = orig_get
    assert sorted(os.listdir(os.path.join(str(tmp_dir), "dev_run"))) == ["data.nc", "metadata.yml"]
    back = retrieve_container(os.path.join(str(tmp_dir), "dev_run"))
    g = np.load(os.path.join(GOLDEN, "simulation.npz"))
    assert back.data["U"].shape == (6, 200) and np.allclose(back.data["t"], np.arange(6) * .5)
    assert np.array_equal(back.data["U"][0], fdict["U"])                    # initial state as given
    assert np.abs(back.data["U"][1:] - g["Theta_ts0_U"]).max() <= 1e-10
    assert np.array_equal(back.data["U"][-1], np.asarray(sim.fields["U"]))
    assert back.metadata["c"] == pars["c"] and back.metadata["periodic"] is False
    assert np.array_equal(c.data["U"], back.data["U"])


def check_model_load_reuses_code_object(backend_cls, tmp_dir):
    """Row f4: ``Model.save`` / ``Model.load`` keep the HIP compiler, and compiling the loaded
    model finds its code object in the on-disk cache: hipcc is not started again."""
    import subprocess
    from triflow_amd import compilers
    m = Model("k * dxxU - c * dxU + s * U", "U", ["k", "c", "s"])      # not a pre-built model
    x = np.linspace(0, 1, 64)
    fd = dict(x=x, U=np.cos(2 * np.pi * x))
    pars = dict(k=.01, c=.3, s=-.2, periodic=False)
    F = m.F(m.fields_template(**fd), pars)                             # builds + caches the code object
    filename = os.path.join(str(tmp_dir), "model.pkl")
    m.save(filename)
    calls = []
    orig_run = subprocess.run

    def run(cmd, *a, **k):
        calls.append(list(cmd))
        return orig_run(cmd, *a, **k)
    subprocess.run = run
    try:
        m2 = Model.load(filename)
        F2 = m2.F(m2.fields_template(**fd), pars)
        J2 = m2.J(m2.fields_template(**fd), pars)
    finally:
        subprocess.run = orig_run
    assert not [c for c in calls if "--genco" in c], calls           # no kernel compilation
    assert np.array_equal(F, F2)
    assert m2._device is not m._device and J2.shape == (64, 64)


def check_unstable_factorisation_recovers(backend):
    """A plan on which block elimination without pivoting across separators breaks down
    (dispersive scalar equation, 4-node chunks): SuperLU never refuses a non-singular system
    (schemes.py:149, 557), and neither does the library -- the factorisation is redone on
    longer chunks (tf_solver::fallback, counted in `replans`), the solution agrees with SuperLU
    and the solve reports that it was not the plain factorisation of the plan asked for.  Later
    factorisations with such a c go to the longer chunks directly.  With the rescue switched off
    (TRIFLOW_REPLAN=0) the same solve raises instead of returning a wrong solution."""
    import os
    import pytest
    name, N, c = "kdv", 203, 0.1
    m, mo = device_model(name, backend), oracle_model(name)
    fd = corpus.synthetic_fields(name, N, seed=7, periodic=True, length=N * 5e-3)
    pars = corpus.synthetic_pars(name, N, True)
    Jo = mo.J(mo.fields_template(**fd), pars)
    A = sps.identity(N, format="csc") - c * Jo
    rhs = np.random.default_rng(5).standard_normal(N)
    xs = spla.spsolve(A, rhs)
    bad = bound_solver(m, fd, pars, m1=4, m_upper=2)
    bad.eval(0, with_j=True)
    bad.factor(c)
    x = bad.solve(rhs)[0]
    assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max()
    omega, refined = bad.backward_error()
    assert refined and omega < 1e-10, (omega, refined)
    first = bad.counters()["replans"]
    assert first >= 1
    bad.factor(c)                                   # the verdict is remembered: no second breakdown
    x = bad.solve(rhs)[0]
    assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max()
    assert bad.counters()["replans"] == first
    os.environ["TRIFLOW_REPLAN"] = "0"
    try:
        loud = bound_solver(m, fd, pars, m1=4, m_upper=2, nstate=2)      # (another cache key: a new solver)
    finally:
        del os.environ["TRIFLOW_REPLAN"]
    loud.eval(0, with_j=True)
    loud.factor(c)
    with pytest.raises(RuntimeError, match="lost accuracy"):
        loud.solve(rhs)
    good = bound_solver(m, fd, pars)
    good.eval(0, with_j=True)
    good.factor(c)
    x = good.solve(rhs)[0]
    assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max()
    omega, refined = good.backward_error()
    assert refined and omega < 1e-10
    assert good.counters()["replans"] == 0


def check_constant_matrix_reuse(backend):
    """Constant-coefficient linear models (every Jacobian entry node- and state-independent: the
    README's advection-diffusion, the diffusion model of config 2): the step functions keep the
    factorisation while c, the parameters and dx are unchanged (tf_set_constant_jacobian) -- same
    states as factorising in every step (to rounding: the right-hand side takes the solve path
    instead of riding with the factorisation), also across a change of dt, of a parameter and of
    both, for Theta (+ hook), ROS2 and BDF-2; a model with a state-dependent Jacobian does not
    qualify."""
    import os
    from triflow_amd.ensemble import Ensemble
    cases = ((1, "Theta", DEVICE_HOOKS["cfg1"], 200), (2, "Theta", None, 3001), (2, "ROS2", None, 1500),
             (1, "BDF2", DEVICE_HOOKS["cfg1"], 333))
    for cfg, sch, hook, N in cases:
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        out = []
        for reuse in ("1", "0"):
            os.environ["TRIFLOW_REUSE_FACTOR"] = reuse
            try:
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2)
            finally:
                del os.environ["TRIFLOW_REUSE_FACTOR"]
            assert ens.solver.constant_jacobian == (reuse == "1")
            states = []
            kidx = list(m._device.pars).index("k")
            for k in range(20):
                if k == 12:
                    ens.solver.set_param(kidx, 3 * pars["k"])          # the matrix changes: new factorisation
                ens.step(dt if k < 6 or k >= 16 else 0.5 * dt)         # so does c = theta dt, twice
                if k in (5, 11, 15, 19):
                    ens.sync()
                    states.append(ens.state().copy())
            ens.close()
            out.append(states)
        for a, b in zip(*out):
            err = np.abs(a - b).max() / np.abs(b).max()
            assert np.isfinite(a).all() and err <= 1e-12, (cfg, sch, err)
        # the parameter change took effect (the states after it differ from a run without it)
        assert np.abs(out[0][2] - out[0][1]).max() > 0
    name, fd, pars, dt, _ = corpus.config_inputs(3, 300)
    m = device_model(name, backend)
    ens = Ensemble(m, fd["x"], {k: v[None, :] for k, v in fd.items() if k != "x"}, pars, True, scheme="ROS2", nstate=2)
    assert not ens.solver.constant_jacobian
    ens.close()


def check_two_resident_factorisations(backend):
    """The step-doubling controller that the reference wraps around every scheme alternates a
    coarse step m*dt and fine steps dt (schemes.py:33-66).  For a constant-matrix model the solver
    keeps TWO factorisations, keyed by c: after the first trial no step factorises again, whatever
    the order of the two step sizes; a third step size takes the place of the one used longest ago;
    a parameter upload invalidates both.  Same states as factorising in every step."""
    import os
    from triflow_amd.ensemble import Ensemble
    for cfg, sch, hook, N in ((1, "Theta", DEVICE_HOOKS["cfg1"], 200), (2, "ROS2", None, 1500)):
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        pattern = ([10 * dt] + [dt] * 10) * 3 + [0.5 * dt, dt, 10 * dt, 0.5 * dt]
        out, counts = [], []
        for reuse in ("1", "0"):
            os.environ["TRIFLOW_REUSE_FACTOR"] = reuse
            try:
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2)
            finally:
                del os.environ["TRIFLOW_REUSE_FACTOR"]
            marks = []
            for k, h in enumerate(pattern):
                ens.step(h)
                if k in (10, 32, len(pattern) - 1):
                    ens.sync()
                    marks.append((ens.state().copy(), ens.solver.counters()["factorisations"]))
            kidx = list(m._device.pars).index("k")
            ens.solver.set_param(kidx, 2 * pars["k"])
            ens.step(dt)
            ens.step(10 * dt)
            ens.sync()
            marks.append((ens.state().copy(), ens.solver.counters()["factorisations"]))
            ens.close()
            out.append([s for s, _ in marks])
            counts.append([c for _, c in marks])
        for a, b in zip(*out):
            err = np.abs(a - b).max() / np.abs(b).max()
            assert np.isfinite(a).all() and err <= 1e-12, (cfg, sch, err)
        # two factorisations for the three trials; then 0.5 dt replaces 10 dt (used longest ago), dt is
        # still there, 10 dt is made again in the place of 0.5 dt (dt was used after it), and so is the
        # second 0.5 dt; the new parameter costs one factorisation per step size
        assert counts[0] == [2, 2, 5, 7], counts[0]
        assert counts[1][2] == len(pattern), counts[1]


def check_rescue_with_two_factorisations(backend):
    """A constant-matrix model whose plan needs the rescue on longer chunks (linear dispersion,
    u_t = -u_xxx, 4-node chunks) under the step-doubling pattern: both resident factorisations are
    delegated to the one child solver, which holds a single factorisation -- a solve re-delegates
    when the child's belongs to the other step size.  Same states as factorising in every step."""
    import os
    from triflow_amd.ensemble import Ensemble
    compiler = hip_compiler if backend is None else partial(hip_compiler, backend=backend)
    N = 203
    x = np.linspace(0, N * 5e-3, N, endpoint=False)
    U = (1.0 + 0.3 * np.cos(2 * np.pi * x / (N * 5e-3)))[None, :]
    dt = 2e-3
    pattern = ([10 * dt] + [dt] * 10) * 2
    out, replans = [], []
    for reuse in ("1", "0"):
        m = Model("-dxxxU", "U", None, compiler=compiler)
        os.environ["TRIFLOW_REUSE_FACTOR"] = reuse
        try:
            ens = Ensemble(m, x, dict(U=U), dict(periodic=True), True, scheme="Theta", nstate=2, m1=4, m_upper=2)
        finally:
            del os.environ["TRIFLOW_REUSE_FACTOR"]
        assert ens.solver.constant_jacobian == (reuse == "1")
        states = []
        for h in pattern:
            ens.step(h)
            ens.sync()
            states.append(ens.state().copy())
        replans.append(ens.solver.counters()["replans"])
        ens.close()
        out.append(states)
    assert replans[0] >= 1, replans                       # (the plan does need the rescue)
    for k, (a, b) in enumerate(zip(*out)):
        err = np.abs(a - b).max() / np.abs(b).max()
        assert np.isfinite(a).all() and err <= 1e-10, (k, err)


def check_ensemble_restart(backend):
    """Ensemble.restart(): back to the initial state on the device (bench.py uses it to keep long
    runs inside the time range where the film model stays smooth) -- the steps after a restart
    repeat the first ones bit for bit, BDF-2 included (its history starts over)."""
    from triflow_amd.ensemble import Ensemble
    for cfg, sch, hook, N in ((3, "ROS2", None, 3001), (5, "BDF2", DEVICE_HOOKS["cfg5"], 1003),
                              (1, "Theta", DEVICE_HOOKS["cfg1"], 200)):
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=3)
        runs = []
        for _ in range(2):
            for _ in range(7):
                ens.step(dt)
            ens.sync()
            runs.append(ens.state().copy())
            ens.restart()
        ens.close()
        assert np.isfinite(runs[0]).all() and np.array_equal(runs[0], runs[1]), (cfg, sch)
        two = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2)
        try:
            two.restart()
            raised = False
        except RuntimeError:
            raised = True
        two.close()
        assert raised


def check_fused_stage_rhs(backend):
    """Right-hand side of Rosenbrock stages i >= 1: tfk_sweep_f_stage_rhs (F of the stage state and
    J @ sum gamma k from one window pass) against the two-kernel form it replaces
    (tfk_sweep_f_stage + tfk_spmv): the same operations in the same order, so the same bits --
    2, 3 and 4-stage tableaux, periodic and clamped with a Dirichlet hook, ragged chunk lengths."""
    import os
    from triflow_amd.ensemble import Ensemble
    cases = [(3, "ROS2", None, 3001), (3, "ROS3PRL", None, 1777), (3, "RODASPR", None, 2500),
             (1, "ROS3PRw", DEVICE_HOOKS["cfg1"], 203)]
    for cfg, sch, hook, N in cases:
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        out = []
        for fuse in ("1", "0"):
            os.environ["TRIFLOW_FUSE_STAGE"] = fuse
            try:
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2)
            finally:
                del os.environ["TRIFLOW_FUSE_STAGE"]
            for k in range(6):
                ens.step(dt)
            ens.sync()
            out.append(ens.state().copy())
            ens.close()
        assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1]), (cfg, sch)


def check_row_monitor(backend):
    """Every Rosenbrock step measures the backward error of its stage-0 solve at one node per chunk
    (tf_solver::monitor_sampled, a launch nobody waits for).  (i) A healthy factorisation reads rounding
    level, like the explicit check of the same solve.  (ii) A factorisation that
    loses accuracy while no explicit check runs -- a dispersive model on 4-node chunks, checks
    switched off -- is reported by the next synchronising call instead of passing silently."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(3, 3000)
    m = device_model(name, backend)
    fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
    ens = Ensemble(m, fd["x"], fields, pars, True, scheme="ROS2", nstate=2)
    ens.step(dt)
    mon = ens.solver.monitor_error()
    chk, _ = ens.solver.backward_error()
    assert 0 <= mon < 1e-12 and 0 < chk < 1e-12, (mon, chk)
    ens.sync()                                                  # resets the monitor
    assert ens.solver.monitor_error() == 0.0
    ens.close()
    # (ii) KdV on 4-node chunks at c / dx^3 >> 1 (the plan of check_unstable_factorisation_recovers),
    # explicit checks off (refine = -2: the monitor only): the steps run unchecked, the next
    # synchronising call raises
    N = 203
    x = np.linspace(0, N * 5e-3, N, endpoint=False)
    mk = device_model("kdv", backend)
    U = (1.0 + 0.3 * np.cos(2 * np.pi * x / x[-1]))[None, :]
    ens = Ensemble(mk, x, dict(U=U), dict(periodic=True), True, scheme="ROS2", nstate=2, m1=4, m_upper=2,
                   refine=-2)
    for _ in range(3):
        ens.step(0.1)
    assert ens.solver.monitor_error() > 1e-6
    raised = False
    try:
        ens.sync()
    except RuntimeError as ex:
        raised = "lost accuracy" in str(ex)
    assert raised
    # the default plan of the same problem passes the monitor
    ens = Ensemble(mk, x, dict(U=U), dict(periodic=True), True, scheme="ROS2", nstate=2, refine=-2)
    for _ in range(3):
        ens.step(0.1)
    worst = ens.solver.monitor_error()
    ens.sync()
    assert worst < 1e-6, worst


def check_theta_bdf2_monitor(backend):
    """Theta and BDF-2 steps have no J @ v pass to ride on: between two synchronising checks every new
    factorisation is probed at one node of every level-1 chunk (a different one in every step) by a
    launch that nobody waits for (tf_solver::monitor_sampled).  (i) Healthy runs read rounding level,
    also on the state form of a BDF-2 step whose solve leaves U + delta (config 5 with its hook).
    (ii) A factorisation that loses accuracy between two checks -- the dispersive model on 4-node
    chunks, explicit checks switched off -- is reported by the next synchronising call."""
    from triflow_amd.ensemble import Ensemble
    for cfg, sch, N, hook in ((5, "BDF2", 4003, DEVICE_HOOKS["cfg5"]), (5, "Theta", 2001, None), (3, "Theta", 3000, None)):
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        m = device_model(name, backend)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=4, refine=-2)
        worst = 0.0
        for _ in range(12):
            ens.step(dt)
            worst = max(worst, ens.solver.monitor_error())
        ens.sync()
        assert 0.0 <= worst < 1e-11, (cfg, sch, worst)
        if cfg == 5:
            assert worst > 0.0                          # (the probe did measure something)
        ens.close()
    N = 203
    x = np.linspace(0, N * 5e-3, N, endpoint=False)
    mk = device_model("kdv", backend)
    U = (1.0 + 0.3 * np.cos(2 * np.pi * x / x[-1]))[None, :]
    for sch in ("Theta", "BDF2"):
        ens = Ensemble(mk, x, dict(U=U), dict(periodic=True), True, scheme=sch, nstate=4, m1=4, m_upper=2, refine=-2)
        for _ in range(4):
            ens.step(0.1)
        assert ens.solver.monitor_error() > 1e-6, sch
        raised = False
        try:
            ens.sync()
        except RuntimeError as ex:
            raised = "lost accuracy" in str(ex)
        assert raised, sch
        # default mode (explicit checks + the probe in between): the checks are spaced out, the probe is not
        ens = Ensemble(mk, x, dict(U=U), dict(periodic=True), True, scheme=sch, nstate=4, refine=-2)
        for _ in range(4):
            ens.step(0.1)
        worst = ens.solver.monitor_error()
        ens.sync()
        assert worst < 1e-6, (sch, worst)
        ens.close()



def check_ensemble_equals_single_members(backend, N=3000, nsys=3, steps=3, exact=True, **opts):
    """nsys members stepped together in one solver (per-member scalar parameters and
    initial conditions) give, member by member, the bits of nsys separate single-member
    solvers: the batch dimension only adds chunks to the same kernels."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(3, N)
    m = device_model(name, backend)
    c = np.array([0.5, 0.75, 1.0])[:nsys]
    We = np.array([0.005, 0.01, 0.02])[:nsys]
    scale = 1.0 + 0.02 * np.arange(nsys)[:, None]
    fields = {k: fd[k][None, :] * scale for k in ("h", "q", "T")}
    ens = Ensemble(m, fd["x"], fields, dict(pars, c=c, We=We), True, scheme="ROS2", **opts)
    for _ in range(steps):
        ens.step(dt)
    ens.sync()
    batch = ens.state()                                    # [nvar][nsys][N]
    for e in range(nsys):
        one = Ensemble(m, fd["x"], {k: v[e:e + 1] for k, v in fields.items()},
                       dict(pars, c=c[e:e + 1], We=We[e:e + 1]), True, scheme="ROS2", **opts)
        for _ in range(steps):
            one.step(dt)
        one.sync()
        if exact:        # same level plan for the batch and the single member: same bits
            assert np.array_equal(one.state()[:, 0, :], batch[:, e, :]), e
        else:            # plans chosen from the total size differ: same solution to solver accuracy
            ref = one.state()[:, 0, :]
            assert np.abs(ref - batch[:, e, :]).max() <= 1e-10 * np.abs(ref).max(), e
    # and the members differ from each other
    assert not np.array_equal(batch[:, 0, :], batch[:, 1, :])
    # member 1 against the scheme API on the same inputs (its solver may use another level
    # plan, hence a tolerance)
    sch = schemes.ROS2(m)
    f = m.fields_template(x=fd["x"], **{k: v[1] for k, v in fields.items()})
    t = 0.0
    for _ in range(steps):
        t, f = sch(t, f, dt, dict(pars, c=float(c[1]), We=float(We[1])))
    ref = f.uflat.reshape(N, 3).T
    assert np.abs(ref - batch[:, 1, :]).max() <= 1e-9 * np.abs(ref).max()


def drift_against_oracle(backend, cfg, N, sch, nsteps=100, marks=(1, 10, 100)):
    """Relative max-norm difference between the device path and the oracle after the given
    numbers of steps ("100-step drift reported")."""
    if cfg == 1:
        name = "M1_advdiff"
        x = np.linspace(0, 1, N)
        fd, pars, dt = {"x": x, "U": np.cos(2 * np.pi * x * 5)}, dict(c=.03, k=.001, periodic=False), 0.5
        kw_d, kw_o = dict(hook=DEVICE_HOOKS["cfg1"]), dict(hook=corpus.dirichlet_hook_cfg1)
    else:
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
        kw_d = dict(hook=DEVICE_HOOKS["cfg5"]) if cfg == 5 else {}
        kw_o = dict(hook=corpus.dirichlet_hook_cfg5) if cfg == 5 else {}
    m, mo = device_model(name, backend), oracle_model(name)
    pick = lambda mod, mm: {"Theta": mod.Theta, "ROS2": mod.ROS2, "BDF2": mod.BDF2,
                            "RODASPR": lambda q: mod.RODASPR(q, time_stepping=False)}[sch](mm)
    dev, ref = pick(schemes, m), pick(ora, mo)
    f_d, f_o, t, out = m.fields_template(**fd), mo.fields_template(**fd), 0.0, {}
    for k in range(1, nsteps + 1):
        _, f_d = dev(t, f_d, dt, pars, **kw_d)
        t, f_o = ref(t, f_o, dt, pars, **kw_o)
        if k in marks:
            out[k] = np.abs(f_d.uflat - f_o.uflat).max() / np.abs(f_o.uflat).max()
    return out


def check_python_hook_stays_resident(backend):
    """The README's hook is a Python function assigning two nodes (``fields.U[0] = 1``).  Those
    assignments are applied on the device (``tf_poke``): the state is uploaded once and never
    downloaded during the run, with and without the step-doubling wrapper, and the results are
    the reference's."""
    from triflow_amd import _capi
    m = device_model("M1_advdiff", backend)
    _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
    g = np.load(os.path.join(GOLDEN, "simulation.npz"))
    counts = dict(up=0, down=0, poke=0)
    orig = {k: getattr(_capi.DeviceSolver, k) for k in ("set_state", "get_state", "get_state_flat", "poke")}

    def counted(name, key):
        def method(self, *a, **k):
            counts[key] += 1
            return orig[name](self, *a, **k)
        return method
    _capi.DeviceSolver.set_state = counted("set_state", "up")
    _capi.DeviceSolver.get_state = counted("get_state", "down")
    _capi.DeviceSolver.get_state_flat = counted("get_state_flat", "down")
    _capi.DeviceSolver.poke = counted("poke", "poke")
    try:
        for ts in (False, True):
            counts.update(up=0, down=0, poke=0)
            sim = Simulation(m, fdict, pars, dt, hook=corpus.dirichlet_hook_cfg1, tmax=2.5,
                             scheme=schemes.Theta, time_stepping=ts)
            for t, fields in sim:
                pass
            # (step doubling starts its coarse and its first fine step from the same host container)
            assert counts["up"] == (2 if ts else 1) and counts["down"] == 0 and counts["poke"] >= 4, (ts, counts)
            U = fields.uflat
            assert counts["down"] == 1
            assert np.abs(U - g["Theta_ts%i_U" % ts][-1]).max() <= 1e-10, ts
    finally:
        for k, v in orig.items():
            setattr(_capi.DeviceSolver, k, v)


def check_neumann_python_hook(backend):
    """A hook that reads single nodes (zero-gradient ends, ``U[0] = U[1]``) stays on the device
    (``tf_peek`` / ``tf_poke``) and gives what the oracle gives with the same hook."""
    from triflow_amd import _capi

    def hook(t, fields, pars):
        fields.U[0] = fields.U[1]
        fields.U[-1] = fields.U[-2]
        return fields, pars
    m, mo = device_model("M1_advdiff", backend), oracle_model("M1_advdiff")
    _, fdict, pars, dt, _ = corpus.config_inputs(1, 200)
    downloads = []
    orig = _capi.DeviceSolver.get_state
    _capi.DeviceSolver.get_state = lambda self, *a, **k: (downloads.append(1), orig(self, *a, **k))[1]
    try:
        dev, ref = schemes.Theta(m), ora.Theta(mo)
        f_d, f_o, t = m.fields_template(**fdict), mo.fields_template(**fdict), 0.0
        for _ in range(6):
            _, f_d = dev(t, f_d, dt, pars, hook=hook)
            t, f_o = ref(t, f_o, dt, pars, hook=hook)
        assert not downloads
    finally:
        _capi.DeviceSolver.get_state = orig
    assert np.abs(f_d.uflat - f_o.uflat).max() <= 1e-12
