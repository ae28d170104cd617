"""GPU suite (-m gpu): the real HIP path through libtriflow_hip.so on an
MI355X, against the oracle and the reference's golden vectors, plus
size-independent properties at the BASELINE sizes."""
import numpy as np
import pytest
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from oracle import corpus
from oracle.gen_golden import STEP_CASES
from tests import parity_cases as pc
from triflow_amd import schemes
from triflow_amd.tableaux import TABLEAUX

pytestmark = pytest.mark.gpu
HIP = None      # default back end: hipcc + libtriflow_hip.so


def test_native_library_is_the_one_running():
    from triflow_amd import compilers
    lib = compilers.HipBackend().library()
    is_device, ndev = lib.runtime_info()
    assert is_device and ndev >= 1
    assert lib.path.endswith("triflow_amd/lib/libtriflow_hip.so")


@pytest.mark.parametrize("name", sorted(corpus.MODELS))
def test_FJ_golden(name):
    pc.check_FJ_golden(name, HIP)


@pytest.mark.parametrize("name,N", [("M3_film", 30011), ("M1_advdiff", 100003),
                                    ("M5_stiff", 20001)])
def test_FJ_bitexact_ragged(name, N):
    pc.check_FJ_bitexact_large(name, HIP, N)


@pytest.mark.parametrize("name", ["M2_diff", "M1_advdiff", "M3_film", "M5_stiff", "kuramoto",
                                  "kdv", "wave", "upwind2_par", "wide4", "six"])
def test_linear_solve_small(name):
    plans = [dict(m1=4, m_upper=2), dict(m1=7, m_upper=3), dict(m1=32, m_upper=8),
             dict(m1=10 ** 6), dict()]
    # (kdv, kuramoto: dispersion-dominated scalar equations, on which elimination without pivoting
    # across separators loses accuracy with very short chunks -- those plans are rescued by the
    # library on longer chunks, DESIGN.md "solver limits"; every plan has to agree with SuperLU)
    # wide4: fourth derivatives at dx = 5e-3, cond(A) ~ 1e9 for both solvers
    tol = {"wide4": 1e-7}.get(name, 1e-9)
    pc.check_linear_solve(name, HIP, 203, plans, tol=tol)


@pytest.mark.parametrize("name", sorted(corpus.SOLVER_MODELS))
def test_linear_solve_block_sizes(name):
    """Block sizes of the reduced levels that the corpus does not reach: b = mp * nvar = 3
    (tri3), 4 (quad4: 4 variables, pair4: 2 variables with 5-point stencils), 7 (seven) -- each
    its own instantiation of the cooperative kernels; refinement off."""
    pc.check_linear_solve(name, HIP, 203, [dict(refine=0), dict(refine=0, m1=8, m_upper=4)],
                          tol=1e-7 if name == "pair4" else 1e-9)
    pc.check_linear_solve(name, HIP, 5003, [dict(refine=0, m1=8)], tol=1e-6 if name == "pair4" else 1e-8)


def test_linear_solve_b8_four_wavefronts_per_chunk(monkeypatch):
    """b = mp * nvar = 8 (wide4) with the 256-thread form of tfk_cr_factor -- forced at a small
    size, and chosen by the runtime on a level of more than 1024 chunks: a chunk hands 2 * 8 * 17 =
    272 entries to the next level, more than the workgroup has threads (the single-pass
    store lost the last 16 of them); refinement off, so that nothing can mask a wrong factor."""
    monkeypatch.setenv("TRIFLOW_CR_FACTOR_BLOCK", "256")
    pc.check_linear_solve("wide4", HIP, 5003, [dict(refine=0, m1=8)], tol=1e-6)
    monkeypatch.delenv("TRIFLOW_CR_FACTOR_BLOCK")
    pc.check_linear_solve("wide4", HIP, 140003, [dict(refine=0, m1=8)], tol=1e-6)


@pytest.mark.parametrize("name", ["M3_film", "M5_stiff", "wide4", "six", "bivar"])
def test_factorisation_is_accurate_without_refinement(name):
    """The automatic refinement must not be what makes a solve right: with it switched
    off the factorisation alone agrees with SuperLU (this is what catches a miscompiled
    solver kernel, DESIGN.md "compiler notes")."""
    pc.check_linear_solve(name, HIP, 203, [dict(refine=0), dict(refine=0, m1=8, m_upper=4)],
                          tol=1e-7 if name == "wide4" else 1e-9)


@pytest.mark.parametrize("name", ["M2_diff", "M3_film", "M5_stiff", "wide4", "six"])
def test_linear_solve_medium(name):
    pc.check_linear_solve(name, HIP, 20011, [dict(), dict(m1=16, m_upper=4)],
                          tol=1e-6 if name == "wide4" else 1e-8)


@pytest.mark.parametrize("case", STEP_CASES, ids=lambda c: c[0])
def test_steps_golden(case):
    pc.check_steps_golden(case, HIP)


@pytest.mark.parametrize("case", [STEP_CASES[0], STEP_CASES[5]], ids=lambda c: c[0])
def test_steps_golden_python_hook(case):
    pc.check_steps_golden(case, HIP, python_hook=True,
                          only=("Theta1", "ROS2", "RODASPR", "ROS3PRw_adapt"))


def test_tiny_grids():
    pc.check_tiny_grids(HIP)


def test_proportional_entries():
    pc.check_proportional_entries(HIP)


def test_bdf2():
    pc.check_bdf2(HIP)


def test_bdf2_interleaved():
    pc.check_bdf2_interleaved(HIP)


def test_bdf2_against_vode():
    pc.check_bdf2_against_vode(HIP)


def test_simulation_golden():
    pc.check_simulation_golden(HIP)


def test_resident_fields():
    pc.check_resident_fields(HIP)


def test_errors():
    pc.check_errors(HIP)


@pytest.mark.parametrize("scheme", [schemes.ROS2, schemes.ROS3PRL, schemes.ROS3PRw,
                                    schemes.RODASPR, schemes.Theta, schemes.BDF2])
def test_heat_steady_state(scheme):
    pc.check_heat_steady_state(HIP, scheme, dirichlet=False)


@pytest.mark.parametrize("scheme", [schemes.ROS3PRL, schemes.ROS3PRw, schemes.RODASPR])
def test_heat_dirichlet(scheme):
    pc.check_heat_steady_state(HIP, scheme, dirichlet=True)


# ---- BASELINE sizes: properties that do not need the (slow) oracle at 1e6 ----
def _cfg_solver(cfg, N=None, **opts):
    name, fd, pars, dt, sch = corpus.config_inputs(cfg, N)
    m = pc.device_model(name, HIP)
    solver = pc.bound_solver(m, fd, pars, **opts)
    return m, solver, fd, pars, dt


@pytest.mark.parametrize("cfg,gamma", [(2, 1.0), (3, TABLEAUX["ROS2"].gamma[0, 0]), (5, 2. / 3.)])
def test_full_size_solver_residual(cfg, gamma):
    """(I - cJ) x = b at the BASELINE size: the residual computed on the host
    with the downloaded Jacobian must be at rounding level, and solving with
    A @ x_known must give x_known back (round trip)."""
    m, solver, fd, pars, dt = _cfg_solver(cfg)
    N, nvar = fd["x"].size, m._nvar
    solver.eval(0, with_j=True)
    J = m._device.pattern(N, pars["periodic"]).assemble(solver.get_J()[0])
    c = gamma * dt
    A = sps.identity(N * nvar, format="csr") - c * J.tocsr()
    rng = np.random.default_rng(0)
    x_known = rng.standard_normal(N * nvar)
    b = A @ x_known
    solver.factor(c)
    x = solver.solve(b)[0]
    r = np.abs(A @ x - b).max() / np.abs(b).max()
    assert r <= 1e-9, r
    # forward error bounded by cond * eps: a loose, size-independent sanity bound
    assert np.abs(x - x_known).max() <= 1e-4 * np.abs(x_known).max()
    y = solver.matvec(x_known)[0]
    assert np.abs(y - J @ x_known).max() <= 1e-11 * np.abs(J @ x_known).max()
    solver.close()


def test_full_size_sweep_matches_subsampled_oracle():
    """F at N = 1e6 (config 3): every 9973-th window recomputed by the oracle on
    the extracted neighbourhood must agree bit for bit (F is local)."""
    name, fd, pars, dt, _ = corpus.config_inputs(3)
    m = pc.device_model(name, HIP)
    mo = pc.oracle_model(name)
    N = fd["x"].size
    F = m.F(m.fields_template(**fd), pars).reshape(N, 3)
    dx = (fd["x"][-1] - fd["x"][0]) / (N - 1)
    for g in range(5, N - 5, 9973):
        sl = slice(g - 2, g + 3)
        sub = {k: v[sl] for k, v in fd.items()}
        sub["x"] = np.arange(5) * dx            # same dx to the last bit is not
        sub_pars = dict(pars, periodic=False)   # guaranteed: compare through dx
        fo = mo.fields_template(**sub)
        # evaluate the oracle's lambdified F directly on the centre node
        env, _, _, _ = __import__("oracle.numpy_path", fromlist=["x"]).stencil_views(
            mo, sub["x"], *[sub[k] for k in mo._dep_vars], *[sub_pars[k] for k in mo._pars],
            False)
        env["dx"] = dx
        from sympy import lambdify
        from oracle.numpy_path import _lambdify_modules
        f_func = lambdify(mo._symbolic_args, mo.F_array.tolist(), modules=_lambdify_modules())
        vals = f_func(*[env[k] for k in mo._args])
        centre = np.array([np.broadcast_to(v, (5,))[2] for v in vals])
        assert np.array_equal(F[g], centre), g


@pytest.mark.parametrize("cfg", [2, 3])
def test_full_size_step_properties(cfg):
    """One implicit step at N = 1e6: finite, and the periodic models conserve
    the mean of their conservative variable to rounding (dU/dt is a divergence)."""
    name, fd, pars, dt, sch = corpus.config_inputs(cfg)
    m = pc.device_model(name, HIP)
    fields = m.fields_template(**fd)
    scheme = schemes.Theta(m) if sch == "Theta" else schemes.ROS2(m)
    t, new = scheme(0.0, fields, dt, pars)
    key = "U" if cfg == 2 else "h"
    before, after = np.asarray(fields[key]), np.asarray(new[key])
    assert np.isfinite(after).all()
    drift = abs(after.mean() - before.mean())
    assert drift <= 1e-11 * max(1.0, abs(before).max()), drift
    assert not np.array_equal(before, after)


def test_full_size_step_config5():
    """BASELINE config 5 at its full size (M5, N = 4e6, clamped, Dirichlet hook A[0] = A[-1] = 1,
    dt = 1e-3): the backward-Euler start step and one two-step BDF-2 step.  Properties that do
    not need the oracle at that size: finite, boundary values exact, the factorisation's
    componentwise backward error at rounding level with no refinement, and -- the scheme being
    linearly implicit -- the defining equation of each step holds for the result:
        (I - c J(U_n)) (U_{n+1} - U_n) = rhs        checked with the device J @ v product."""
    name, fd, pars, dt, sch = corpus.config_inputs(5)
    assert sch == "BDF2" and fd["x"].size == 4 * 10 ** 6
    m = pc.device_model(name, HIP)
    hook = pc.DEVICE_HOOKS["cfg5"]
    scheme = schemes.BDF2(m)
    f0 = m.fields_template(**fd)
    t1, f1 = scheme(0.0, f0, dt, pars, hook=hook)          # backward-Euler form
    solver = f1._device_backing().stepper.solver
    om1, refined1 = solver.backward_error()
    t2, f2 = scheme(t1, f1, dt, pars, hook=hook)           # two-step form
    om2, refined2 = solver.backward_error()
    print("config 5 full size: backward error %.1e (BE step), %.1e (BDF-2 step)" % (om1, om2))
    assert om1 < 1e-10 and om2 < 1e-10 and not refined1 and not refined2, (om1, om2)
    u0 = np.array([np.asarray(fd[k]) for k in m._dep_vars])
    u0[0, 0] = u0[0, -1] = 1.0                              # the hook at t = 0
    u1 = np.array([np.asarray(f1[k]) for k in m._dep_vars])
    u2 = np.array([np.asarray(f2[k]) for k in m._dep_vars])
    for u in (u1, u2):
        assert np.isfinite(u).all()
        assert u[0, 0] == 1.0 and u[0, -1] == 1.0           # boundary values exact
    assert not np.array_equal(u1, u2)
    # defining equation of the second step, with the J the step itself evaluated (still resident:
    # the J @ v kernel of the same solver) and F evaluated once more at the step's input (the fused
    # sweep of the step leaves F inside its right-hand side; an F-only evaluation keeps J)
    N, nvar = fd["x"].size, m._nvar
    flat = lambda u: u.T.reshape(-1)
    F1 = np.asarray(m.F(f1, pars)).reshape(-1)
    d2 = flat(u2) - flat(u1)
    rhs = (flat(u1) - flat(u0)) / 3.0 + (2.0 / 3.0) * dt * F1
    res = (d2 - (2.0 / 3.0) * dt * solver.matvec(d2)[0] - rhs).reshape(N, nvar)
    # the two hooked nodes are overwritten after the solve, which also enters their
    # neighbours' rows through J: leave nodes 0, 1, N-2, N-1 out
    scale = np.abs(rhs).max()
    worst = np.abs(res[2:-2]).max() / scale
    print("config 5 full size: BDF-2 defining-equation residual %.1e (relative)" % worst)
    assert worst <= 1e-10, worst


def test_config4_shard_full_size():
    """BASELINE config 4 per GPU: 8 members x N = 1e6 of the film model in one solver, with the
    sweep's parameter table (c = 0.5 + m/64, We = 0.005 (1 + m mod 8), phase 2 pi m/64;
    rank 3's members of the 64: m = 3, 11, ..., 59).  Finite, mean of h conserved per member,
    and one member equals a single-member solver on the same inputs."""
    import importlib.util, os
    from triflow_amd.ensemble import Ensemble, shard_members
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = bench.member_table(64, None)[shard_members(64, 3, 8)]
    assert table.shape == (8, 3)
    name, x, fields, pars, dt, scheme = bench.build_problem(3, None, table)
    assert x.size == 10 ** 6 and scheme == "ROS2" and fields["h"].shape == (8, 10 ** 6)
    m = pc.device_model(name, HIP)
    ens = Ensemble(m, x, fields, pars, True, scheme=scheme, nstate=2)
    nsteps = 2
    for _ in range(nsteps):
        ens.step(dt)
    ens.sync()
    out = ens.state()                                        # [nvar][nsys][N]
    ens.close()
    assert np.isfinite(out).all()
    for e in range(8):
        drift = abs(out[0, e].mean() - fields["h"][e].mean())
        assert drift <= 1e-11, (e, drift)
    assert not np.array_equal(out[:, 0], out[:, 1])
    e = 3
    member = lambda: Ensemble(m, x, {k: v[e:e + 1] for k, v in fields.items()},
                              dict(pars, c=pars["c"][e:e + 1], We=pars["We"][e:e + 1]), True,
                              scheme=scheme, nstate=2)

    def run(ens1):
        for _ in range(nsteps):
            ens1.step(dt)
        ens1.sync()
        out1 = ens1.state()[:, 0, :]
        ens1.close()
        return out1
    # (a) the level plan a single member gets by default is the batch's plan (chunk counts do not
    # depend on the number of members): the batch dimension only adds chunks to the same
    # kernels, member 3 comes out bit for bit.  (Both run their re-elimination walks in twisted form
    # since round 3: the batch's 250 000 chunks too, because the two launches are one there,
    # tfk_l1_fwd2_backsub.  The one-sided form of the same solve: rounding, bound as in (b).)
    assert np.array_equal(run(member()), out[:, e, :])
    ref = out[:, e, :]
    os.environ["TRIFLOW_L1_TWIST"] = "0"
    try:
        single = member()
    finally:
        del os.environ["TRIFLOW_L1_TWIST"]
    err = np.abs(run(single) - ref).max() / np.abs(ref).max()
    print("config 4 shard: member %d vs single-member solver (one-sided walks) %.1e" % (e, err))
    assert err <= 3e-8, err
    # (b) another level plan (chunk walks instead of cyclic reduction above 5000 nodes):
    # another elimination order of a matrix with cond(I - gamma dt J) = 2e10,
    # so the two backward-stable solves differ by cond * eps; measured 2.8e-10, bound 100 x
    os.environ["TRIFLOW_CR_MAX_NODES"] = "5000"
    try:
        ref = run(member())
    finally:
        del os.environ["TRIFLOW_CR_MAX_NODES"]
    err = np.abs(ref - out[:, e, :]).max() / np.abs(ref).max()
    print("config 4 shard: member %d vs single-member solver (walk plan) %.1e" % (e, err))
    assert err <= 3e-8, err


def test_step_doubling_device_norm():
    pc.check_step_doubling_device_norm(HIP)


def test_fused_step_doubling():
    pc.check_fused_step_doubling(HIP)


def test_graph_replay_equals_eager():
    """Fixed steps of small grids are captured into HIP graphs and replayed (tf_solver::run_graphed):
    the same launches, so the same bits as issuing them one by one -- Theta with a Dirichlet hook,
    ROS2 and RODASPR (fixed step) -- and a change of dt or of the boundary values is honoured."""
    import os
    from triflow_amd.ensemble import Ensemble
    cases = [(1, "Theta", pc.DEVICE_HOOKS["cfg1"]), (3, "ROS2", None), (3, "RODASPR", None)]
    for cfg, sch, hook in cases:
        name, fd, pars, dt, _ = corpus.config_inputs(cfg, 200 if cfg == 1 else 3000)
        m = pc.device_model(name, HIP)
        fields = {k: v[None, :] for k, v in fd.items() if k != "x"}
        out = []
        for graphs in ("1", "0"):
            os.environ["TRIFLOW_GRAPHS"] = graphs
            try:
                ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2)
            finally:
                del os.environ["TRIFLOW_GRAPHS"]
            for k in range(30):
                if k == 12:
                    # a state from outside: the slot no longer holds what a step left with the hook
                    # applied, the next step starts from a hooked copy again (another string of launches)
                    ens.solver.set_state(ens.cur, ens.state() * 1.001)
                ens.step(dt if k < 20 else 0.5 * dt)          # the last ten with another step size
            ens.sync()
            out.append(ens.state().copy())
            ens.close()
        assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1]), (cfg, sch)


def test_time_dependent_hook():
    pc.check_time_dependent_hook(HIP)


# ---- BASELINE configurations against the oracle at sizes it still finishes in seconds ----
@pytest.mark.parametrize("cfg,N,nsteps,tol", [(2, 10 ** 6, 2, 1e-7), (3, 2 * 10 ** 5, 2, 2.5e-10),
                                              (5, 4 * 10 ** 5, 3, 2e-11),
                                              (3, 10 ** 6, 2, 2e-8), (5, 2 * 10 ** 6, 3, 4e-10)])
def test_config_steps_vs_oracle(cfg, N, nsteps, tol):
    """Configs 2 (full size), 3 and 5 (1/5 and 1/10 size, same dx scaling rules as
    corpus.config_inputs): the configured scheme on the device against the oracle
    (reference algorithm + SuperLU); config 3 also at its full size (the metric's configuration)
    and config 5 at half of it.  Tolerances = 100 x the measured difference (1.4e-9, 2.6e-12,
    1.7e-13, 1.6e-10, 3.4e-12 on MI355X; the measured value is printed and shown on failure).
    The differences are cond(A)*eps of the two direct solvers: config 2 at N = 1e6 has
    cond(I - dt J) = 4e7."""
    from oracle import numpy_path as ora
    name, fd, pars, dt, sch = corpus.config_inputs(cfg, N)
    m, mo = pc.device_model(name, HIP), pc.oracle_model(name)
    dev = {"Theta": schemes.Theta, "ROS2": schemes.ROS2, "BDF2": schemes.BDF2}[sch](m)
    ref = {"Theta": ora.Theta, "ROS2": ora.ROS2, "BDF2": ora.BDF2}[sch](mo)
    kw_d = dict(hook=pc.DEVICE_HOOKS["cfg5"]) if cfg == 5 else {}
    kw_o = dict(hook=corpus.dirichlet_hook_cfg5) if cfg == 5 else {}
    f_d, f_o = m.fields_template(**fd), mo.fields_template(**fd)
    t = 0.0
    for k in range(nsteps):
        _, f_d = dev(t, f_d, dt, pars, **kw_d)
        t, f_o = ref(t, f_o, dt, pars, **kw_o)
    omega, refined = f_d._device_backing().stepper.solver.backward_error()
    u_d, u_o = f_d.uflat, f_o.uflat
    err = np.abs(u_d - u_o).max() / np.abs(u_o).max()
    print("config %d N=%d: rel err vs oracle %.2e, backward error %.1e" % (cfg, N, err, omega))
    assert err <= tol, "config %d: rel err vs oracle %.3e > %.1e" % (cfg, err, tol)
    assert omega < 1e-10 and not refined, (omega, refined)


@pytest.mark.parametrize("name", sorted(pc.NOTEBOOK_CASES))
def test_notebook_models(name):
    pc.check_notebook_model(name, HIP)


def test_simulation_stays_resident():
    pc.check_simulation_stays_resident(HIP)


def test_container_on_device_fields(tmp_path):
    pc.check_container_on_device_fields(HIP, tmp_path)


def test_model_load_reuses_code_object(tmp_path):
    pc.check_model_load_reuses_code_object(HIP, tmp_path)


def test_row_monitor():
    pc.check_row_monitor(HIP)


def test_rescue_with_two_factorisations():
    pc.check_rescue_with_two_factorisations(HIP)


def test_two_resident_factorisations():
    pc.check_two_resident_factorisations(HIP)


def test_constant_matrix_reuse():
    pc.check_constant_matrix_reuse(HIP)


@pytest.mark.parametrize("cfg,sch,N,nsys", [(3, "ROS2", 3001, 2), (5, "BDF2", 2003, 1), (2, "Theta", 3001, 1),
                                            (1, "Theta", 200, 1), (3, "RODASPR", 40000, 1)])
def test_walks_assemble_the_separator_rows(cfg, sch, N, nsys, monkeypatch):
    """Below a cyclic-reduction level the level-1 walks write their halves of the separator rows
    themselves (tf_asm_side) instead of leaving tips for tfk_l1_asm_mat / _rhs: the same products,
    the diagonal block summed in two parts -- states equal to rounding (periodic and clamped, hook,
    two members, scalar blocks with row exchanges, a plan with several cyclic-reduction levels)."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
    m = pc.device_model(name, HIP)
    fields = {k: np.repeat(v[None, :], nsys, axis=0) * (1 + 0.01 * np.arange(nsys))[:, None]
              for k, v in fd.items() if k != "x"}
    hook = pc.DEVICE_HOOKS["cfg5"] if cfg == 5 else (pc.DEVICE_HOOKS["cfg1"] if cfg == 1 else None)
    out = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("TRIFLOW_L1_FUSE_ASM", fuse)
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2, refine=0)
        for _ in range(4):
            ens.step(dt)
        ens.sync()
        out.append(ens.state().copy())
        ens.close()
    err = np.abs(out[0] - out[1]).max() / np.abs(out[1]).max()
    print("walk-assembled separator rows vs assemble kernels, config %d %s: %.1e" % (cfg, sch, err))
    assert np.isfinite(out[0]).all() and err <= 1e-10, err


@pytest.mark.parametrize("name,N,periodic,opt", [("M3_film", 3001, True, None), ("M3_film", 2003, False, None),
                                                 ("bivar", 1501, False, None), ("M3_film", 3001, True, "-O1")])
def test_split_factorisation_walk_equals_one_wavefront(name, N, periodic, opt, monkeypatch):
    """tfk_l1_factor* by two wavefronts per 64 chunks and direction (one eliminates the band, the
    other carries the spike columns and the first right-hand side with the pivot blocks published
    in LDS; TF_L1_SPLIT_MODEL) does the arithmetic of the one-wavefront walk: the same products in
    the same order -- solutions equal to the last bits, with the stored spike response and with
    the second elimination, and in steps whose first solve rides with the factorisation.  The grids
    have ragged chunks (lanes of a wavefront with trip counts that differ by one: the two wavefronts
    of a walk meet at one barrier per pivot, and the per-lane slots keep their parity whatever the
    compiler makes of the loops: also at -O1)."""
    from triflow_amd import compilers
    from triflow_amd.ensemble import Ensemble
    if opt:
        monkeypatch.setattr(compilers, "HIPCC_FLAGS", [opt] + [f for f in compilers.HIPCC_FLAGS if not f.startswith("-O")])
    rng = np.random.default_rng(17)
    fd = corpus.synthetic_fields(name, N, seed=5, periodic=periodic, length=N * 5e-3)
    pars = corpus.synthetic_pars(name, N, periodic)
    rhs = rng.standard_normal(N * len(corpus.model_args(name)[1]))
    results = {}
    for form in ("split", "one"):
        if form == "one":
            monkeypatch.setattr(compilers, "HIPCC_FLAGS", compilers.HIPCC_FLAGS + ["-DTF_L1_SPLIT=0"])
        m = pc.device_model(name, HIP)
        xs = []
        for respike in ("0", "1"):
            monkeypatch.setenv("TRIFLOW_L1_RESPIKE", respike)
            solver = pc.bound_solver(m, fd, pars)
            assert solver.kernel_block("tfk_l1_factor_rhs") == (128 if form == "split" else 64)
            solver.eval(0, with_j=True)
            solver.factor(0.01)
            xs.append(solver.solve(rhs)[0].copy())
            ens = Ensemble(m, fd["x"], {k: v for k, v in fd.items() if k != "x"}, pars, periodic,
                           scheme="ROS2", nstate=2, refine=0)
            for _ in range(3):
                ens.step(1e-3)
            ens.sync()
            xs.append(ens.state().copy())
            ens.close()
        results[form] = xs
    for a, b in zip(results["split"], results["one"]):
        err = np.abs(a - b).max() / np.abs(b).max()
        assert np.isfinite(a).all() and err <= 1e-13, err


def test_ensemble_restart():
    pc.check_ensemble_restart(HIP)


def test_bdf2_history_in_place():
    pc.check_bdf2_history_in_place(HIP)


def test_bdf2_history_is_the_hooked_state():
    pc.check_bdf2_history_is_the_hooked_state(HIP)


def test_hook_input_in_place():
    pc.check_hook_input_in_place(HIP)


def test_respike():
    pc.check_respike(HIP)


@pytest.mark.parametrize("m1", [32, 13, 9])
def test_fused_level1_backsub_equals_two_launches(m1, monkeypatch):
    """tfk_l1_fwd2_backsub (second elimination and back-substitution of the twisted form in one
    launch, y in LDS) does the arithmetic of tfk_l1_fwd2 + tfk_l1_backsub_u: the same bits, on
    chunk lengths that are split in two, and on a level that mixes split and one-sided chunks."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(3, 3001)
    m = pc.device_model(name, HIP)
    fields = {k: np.repeat(v[None, :], 2, axis=0) * (1 + 0.01 * np.arange(2))[:, None]
              for k, v in fd.items() if k != "x"}
    out = []
    monkeypatch.setenv("TRIFLOW_L1_RESPIKE", "1")
    for fuse in ("1", "0"):
        monkeypatch.setenv("TRIFLOW_L1_FUSE_BACKSUB", fuse)
        ens = Ensemble(m, fd["x"], fields, pars, True, scheme="RODASPR", nstate=2, m1=m1)
        for _ in range(4):
            ens.step(dt)
        ens.sync()
        out.append(ens.state().copy())
        ens.close()
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])


@pytest.mark.parametrize("cfg,sch,N,nsys,m1", [(3, "ROS2", 3001, 2, 32), (3, "ROS2", 2003, 1, 13),
                                                 (5, "BDF2", 3001, 2, 32), (3, "RODASPR", 3001, 1, 32)])
def test_state_update_inside_the_back_substitution(cfg, sch, N, nsys, m1, monkeypatch):
    """The last solve of a fixed step of one or two stages leaves the new state instead of its
    solution (tfk_l1_fwd2_backsub with TfLevelArgs::upd_*: base + (b0 k0 + b1 x), the operations of
    the vector kernel in their order): the same bits as solve + tfk_vec, with a hook, with two
    members, on chunks that are split in two and on one-sided ones; and the vector kernel is gone
    from those steps (schemes with more stages keep it)."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
    m = pc.device_model(name, HIP)
    fields = {k: np.repeat(v[None, :], nsys, axis=0) * (1 + 0.01 * np.arange(nsys))[:, None]
              for k, v in fd.items() if k != "x"}
    hook = pc.DEVICE_HOOKS["cfg5"] if cfg == 5 else None
    monkeypatch.setenv("TRIFLOW_L1_RESPIKE", "1")
    out, vec_launches = [], []
    for fuse in ("1", "0"):
        monkeypatch.setenv("TRIFLOW_FUSE_UPDATE", fuse)
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2, m1=m1, refine=0)
        ens.solver.timing(True)
        for _ in range(5):
            ens.step(dt)
        ens.sync()
        out.append(ens.state().copy())
        vec_launches.append(ens.solver.timing_report().get("tfk_vec", (0.0, 0))[1])
        ens.close()
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
    assert vec_launches[1] == 5 and vec_launches[0] == (5 if sch == "RODASPR" else 0), vec_launches


@pytest.mark.parametrize("cfg,sch,N,nsys,m1", [(3, "ROS2", 70003, 1, 32), (3, "RODASPR", 40009, 2, 13),
                                                 (5, "BDF2", 50021, 1, 16), (5, "Theta", 33013, 3, 11),
                                                 (3, "ROS2", 9001, 1, 4)])
def test_level_two_inside_the_level_one_launches(cfg, sch, N, nsys, m1, monkeypatch):
    """3 <= b <= 6 and a plan of four or more levels: tfk_l1_solve_cr / tfk_l1_fwd2_backsub_cr run level 2's
    cyclic reduction in the workgroups that walk the level-1 chunks around its nodes -- the bodies of
    tfk_l1_solve + tfk_cr_fwd and of tfk_cr_bwd + tfk_l1_fwd2_backsub, the same bits -- with one and several
    members, periodic and clamped with a hook, level-2 chunk counts that are no multiple of four."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(cfg, N)
    m = pc.device_model(name, HIP)
    fields = {k: np.repeat(v[None, :], nsys, axis=0) * (1 + 0.01 * np.arange(nsys))[:, None]
              for k, v in fd.items() if k != "x"}
    hook = pc.DEVICE_HOOKS["cfg5"] if cfg == 5 else None
    monkeypatch.setenv("TRIFLOW_L1_RESPIKE", "1")
    out, reps = [], []
    for fuse in ("1", "0"):
        monkeypatch.setenv("TRIFLOW_L1CR_FUSE", fuse)
        ens = Ensemble(m, fd["x"], fields, pars, bool(pars["periodic"]), scheme=sch, hook=hook, nstate=2, m1=m1, refine=0)
        assert len(ens.solver.describe()["chunks"]) >= 4, ens.solver.describe()
        ens.solver.timing(True)
        for _ in range(4):
            ens.step(dt)
        ens.sync()
        out.append(ens.state().copy())
        reps.append(ens.solver.timing_report())
        ens.close()
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
    assert "tfk_l1_fwd2_backsub_cr" in reps[0] and "tfk_l1_fwd2_backsub" not in reps[0], sorted(reps[0])
    assert "tfk_l1_fwd2_backsub_cr" not in reps[1] and "tfk_l1_fwd2_backsub" in reps[1], sorted(reps[1])
    assert reps[0]["tfk_cr_bwd"][1] < reps[1]["tfk_cr_bwd"][1]       # (one level fewer in the reduced-level launches)
    if "tfk_l1_solve" in reps[1]:                  # (a step with a solve that does not ride with the factorisation)
        assert "tfk_l1_solve_cr" in reps[0] and "tfk_l1_solve" not in reps[0], sorted(reps[0])
        assert "tfk_l1_solve_cr" not in reps[1] and reps[0].get("tfk_cr_fwd", (0, 0))[1] < reps[1]["tfk_cr_fwd"][1]
    else:
        assert sch in ("BDF2", "Theta")


def test_fused_stage_rhs():
    pc.check_fused_stage_rhs(HIP)


def test_unstable_factorisation_recovers():
    pc.check_unstable_factorisation_recovers(HIP)


def test_ensemble_equals_single_members():
    pc.check_ensemble_equals_single_members(HIP, N=30000, m1=32)
    # default plans: chunk length and reduced-level kernels follow the total size of the batch
    pc.check_ensemble_equals_single_members(HIP, N=30000, exact=False)
    pc.check_ensemble_equals_single_members(HIP, N=3000, nsys=2, m1=8, m_upper=3)


@pytest.mark.parametrize("cfg,N,sch,tol1,tol100", [
    (1, 200, "Theta", 7e-14, 1e-12), (2, 20000, "Theta", 7e-11, 7e-9), (3, 20000, "ROS2", 4e-13, 4e-12),
    (3, 20000, "RODASPR", 4e-13, 8e-12), (5, 20000, "BDF2", 3e-14, 1e-12)])
def test_hundred_step_drift(cfg, N, sch, tol1, tol100):
    """Device path vs oracle after 1 / 100 steps; tolerances = 100 x the values measured on MI355X
    (7e-16 / 1e-14, 7e-13 / 7e-11, 4e-15 / 4e-14, 4e-15 / 8e-14, 1e-16 / 1e-14; not below 100 eps);
    tools/gpu_drift.py prints the table."""
    d = pc.drift_against_oracle(HIP, cfg, N, sch)
    msg = "config %d N=%d %s: %.1e / %.1e / %.1e after 1 / 10 / 100 steps" % (cfg, N, sch, d[1], d[10], d[100])
    print(msg)
    assert d[1] <= tol1 and d[100] <= tol100, msg


@pytest.mark.parametrize("script,args", [("advection_diffusion.py", []), ("film_rosenbrock.py", ["20000"]),
                                          ("parameter_sweep.py", ["20000", "4"])])
def test_examples_run(script, args):
    """The scripts under examples/ (the reference README's example among them) run as they are."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "examples", script), *args], cwd=root,
                         env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    if script == "advection_diffusion.py":
        assert "t: 2.5" in res.stdout and res.stdout.count("iteration") == 5, res.stdout


def test_bench_process_group_one_rank():
    """The N > 1 leg of bench.py -- RCCL process group with the device bound, parameter-table
    broadcast, barriers around the timed blocks, all-gather of the block times and of the device
    identities -- executed with the one rank a one-GPU box allows (TRIFLOW_BENCH_FORCE_DIST=1)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRIFLOW_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TRIFLOW_BENCH_CONFIG4="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--nodes", "20000",
                          "--steps", "5", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline", "--plain"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["backend"].startswith("nccl") and line["ranks_seen"] == 1 and len(line["devices_seen"]) == 1
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["steps_per_s_per_rank"]) == 1
    assert abs(line["steps_per_s_per_rank"][0] - line["value"]) <= 1e-2 * line["value"]
    # the config-4 block of the 8-rank line (8 members per rank), rehearsed with this one rank
    c4 = line["config4"]
    assert c4["members"] == 8 and c4["members_per_rank"] == [8] and c4["member_steps_per_s"] > 0
    assert len(c4["per_rank"]) == 1


def test_bench_two_ranks_rehearsal():
    """The N > 1 path of bench.py with two real ranks on this one GPU (Gloo process group, both on
    device 0: RCCL refuses ranks that share a card): member sharding by rank, the broadcast of the
    table, barriers, the max over ranks per block, the self-description of the line, and the
    config-4 block (8 members per rank) -- launched the way the driver launches it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRIFLOW_BENCH_BACKEND="gloo", TRIFLOW_BENCH_CONFIG4="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--nodes", "100000", "--steps", "5", "--warmup", "2", "--repeats", "3",
                          "--no-cpu-baseline", "--plain"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["members_per_rank"] == [1, 1]
    assert line["backend"].startswith("gloo") and len(line["steps_per_s_per_rank"]) == 2
    assert line["value"] > 0 and line["scaling"] == "weak"
    # value = both members over the slowest rank's block time: not above the sum of the ranks' own rates
    assert line["value"] <= 1.001 * sum(line["steps_per_s_per_rank"])
    c4 = line["config4"]
    assert c4["members"] == 16 and c4["members_per_rank"] == [8, 8] and len(c4["per_rank"]) == 2


def test_bench_line_carries_its_parity():
    """bench.py holds the device state of its own run to the oracle state of its cpu_baseline leg
    (same inputs, same steps) and fails above the bound: here at a size the oracle does in a second."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--nodes", "20000",
                          "--steps", "5", "--warmup", "2", "--repeats", "3", "--plain"],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    par = line["parity"]
    assert par["steps"] == 3 and par["rel_err"] <= par["bound"] and par["backward_error"] < 1e-10, par
    assert line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0


def test_python_hook_stays_resident():
    pc.check_python_hook_stays_resident(HIP)


def test_neumann_python_hook():
    pc.check_neumann_python_hook(HIP)


def test_adaptive_landing_reuse():
    pc.check_adaptive_landing_reuse(HIP)


@pytest.mark.parametrize("name,N", [("M2_diff", 20011), ("kdv", 70003), ("burgers", 5003), ("bivar", 9001),
                                    ("M1_advdiff", 300007), ("M2_diff", 1500007)])
def test_scalar_solve_in_two_launches(name, N, monkeypatch):
    """b = mp * nvar <= 2 with the plan [level 1 | 256-node chunks | one chunk]: tfk_s_fwd / tfk_s_bwd run
    the kernels' bodies of the six-launch solve in two launches -- the same operations, the same bits;
    against SuperLU as well; periodic and clamped; a solve that rides with the factorisation and later ones.
    (N = 1.5e6: a last level of one 367-node chunk, TF_CRS_TOPLEN.)"""
    m, mo = pc.device_model(name, HIP), pc.oracle_model(name)
    rng = np.random.default_rng(11)
    for periodic in (True, False):
        fd = corpus.synthetic_fields(name, N, seed=7, periodic=periodic, length=N * 5e-3)
        pars = corpus.synthetic_pars(name, N, periodic)
        n = N * m._nvar
        rhs = rng.standard_normal(n)
        out = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("TRIFLOW_S_FUSE", fuse)
            solver = pc.bound_solver(m, fd, pars, refine=0)
            assert len(solver.describe()["chunks"]) == 3, solver.describe()
            solver.eval(0, with_j=True)
            solver.factor(0.01)
            out[fuse] = [solver.solve(rhs)[0], solver.solve(rhs[::-1].copy())[0]]
            m._device.release()          # (the per-shape solver cache: the switch is read when a solver is made)
        for a, b in zip(out["1"], out["0"]):
            assert np.isfinite(a).all() and np.array_equal(a, b), (name, periodic, np.abs(a - b).max())
        Jo = mo.J(mo.fields_template(**fd), pars)
        xs = spla.spsolve(sps.identity(n, format="csc") - 0.01 * Jo, rhs)
        err = np.abs(out["1"][0] - xs).max() / np.abs(xs).max()
        # (refinement is off here: the dispersion-dominated model keeps what its elimination loses, ~1e-7)
        assert err <= (1e-6 if name == "kdv" else 1e-8), (name, periodic, err)
    # the launches a Theta step of the constant-matrix model makes once its factorisation is kept
    if name == "M2_diff":
        monkeypatch.setenv("TRIFLOW_S_FUSE", "1")
        _, fdc, parsc, dt, _ = corpus.config_inputs(2, 40000)
        sch = schemes.Theta(m)
        f, t = m.fields_template(**fdc), 0.0
        for _ in range(3):
            t, f = sch(t, f, dt, parsc)
        solver = f._device_backing().stepper.solver
        solver.timing(True); solver.timing_reset()
        t, f = sch(t, f, dt, parsc)
        solver.sync()
        rep = solver.timing_report()
        solver.timing(False)
        assert set(rep) == {"tfk_sweep_fj_theta", "tfk_s_fwd", "tfk_s_bwd"}, rep


def test_theta_bdf2_monitor():
    pc.check_theta_bdf2_monitor(HIP)


def test_scalar_solve_in_two_launches_ensemble(monkeypatch):
    """... with several systems in one solver (a workgroup per level-2 chunk and system, one counter per
    system): three members of the diffusion model with different coefficients, Theta and ROS2 steps, the
    two-launch solve against the six launches, bit for bit."""
    from triflow_amd.ensemble import Ensemble
    name, fd, pars, dt, _ = corpus.config_inputs(2, 20011)
    m = pc.device_model(name, HIP)
    fields = {"U": np.stack([fd["U"] * (1 + 0.1 * e) for e in range(3)])}
    p3 = dict(pars, k=np.array([1e-3, 2e-3, 5e-4]))
    for sch in ("Theta", "ROS2"):
        out = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("TRIFLOW_S_FUSE", fuse)
            ens = Ensemble(m, fd["x"], fields, p3, True, scheme=sch, nstate=2, refine=0)
            assert len(ens.solver.describe()["chunks"]) == 3
            for _ in range(4):
                ens.step(dt)
            ens.sync()
            out[fuse] = ens.state().copy()
            ens.close()
        assert np.isfinite(out["1"]).all() and np.array_equal(out["1"], out["0"]), sch
        assert np.abs(out["1"][0, 0] - out["1"][0, 1]).max() > 0        # (the members do differ)
