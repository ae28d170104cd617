"""CPU suite: the kernel bodies of csrc/tf_kernels.h and the whole host stack
(runtime C ABI, ctypes binding, compiler plugin, schemes, Simulation) executed
through the test-only host emulation (tests/emu) and compared with the oracle
and the reference's golden vectors.  The GPU suite (test_gpu_parity.py) runs
the same checks through the real HIP path."""
import pytest

from oracle.gen_golden import STEP_CASES
from tests import parity_cases as pc
from tests.emu.build_emu import EmuBackend
from triflow_amd import schemes


@pytest.fixture(scope="module")
def backend():
    return EmuBackend()


FJ_MODELS = ["M1_advdiff", "M2_diff", "heat_nopar", "bivar", "helper", "helper_d",
             "upwind1_const", "upwind2_par", "upwind3_par", "upwind2_state", "burgers", "kdv",
             "kuramoto", "wave", "nonlin", "M3_film", "M5_stiff", "wide4", "six", "euler_const"]


@pytest.mark.parametrize("name", FJ_MODELS)
def test_FJ_golden(name, backend):
    pc.check_FJ_golden(name, backend)


@pytest.mark.parametrize("name,N", [("M3_film", 301), ("M1_advdiff", 1000)])
def test_FJ_ragged_chunks(name, N, backend):
    pc.check_FJ_bitexact_large(name, backend, N)


@pytest.mark.parametrize("name", ["M2_diff", "M3_film", "M5_stiff", "kuramoto", "kdv", "wide4", "six"])
def test_linear_solve(name, backend):
    plans = [dict(m1=4, m_upper=2), dict(m1=7, m_upper=3), dict(m1=32, m_upper=8),
             dict(m1=10 ** 6)]
    # wide4: fourth derivatives at dx = 5e-3, cond(A) ~ 1e9 for both solvers
    pc.check_linear_solve(name, backend, 203, plans, tol=1e-7 if name == "wide4" else 1e-9)


@pytest.mark.parametrize("name", ["M2_diff", "M3_film", "M5_stiff", "six"])
def test_linear_solve_walk_levels(name, backend, monkeypatch):
    """The reduced levels as chunk walks (tfk_bt_*: what block sizes above 8 run on the GPU, and
    what TRIFLOW_REDUCED=walk selects); the other tests of this file run them as cyclic reduction
    (tf_crs.h: one host "thread" per chunk)."""
    monkeypatch.setenv("TRIFLOW_REDUCED", "walk")
    plans = [dict(m1=4, m_upper=2), dict(m1=7, m_upper=3), dict(m1=32, m_upper=8)]
    pc.check_linear_solve(name, backend, 203, plans, tol=1e-9)


@pytest.mark.parametrize("case", [c for c in pc.STEP_CASES if c[0] in
                                  ("cfg1", "film_per", "stiff_clamp")],
                         ids=lambda c: c[0])
def test_steps_golden(case, backend):
    pc.check_steps_golden(case, backend)


def test_steps_golden_python_hook(backend):
    pc.check_steps_golden(STEP_CASES[0], backend, python_hook=True,
                          only=("Theta1", "ROS2", "RODASPR_adapt"))


def test_tiny_grids(backend):
    pc.check_tiny_grids(backend)


def test_proportional_entries(backend):
    pc.check_proportional_entries(backend)


def test_bdf2(backend):
    pc.check_bdf2(backend)


def test_bdf2_interleaved(backend):
    pc.check_bdf2_interleaved(backend)


def test_rescue_with_two_factorisations(backend):
    pc.check_rescue_with_two_factorisations(backend)


def test_two_resident_factorisations(backend):
    pc.check_two_resident_factorisations(backend)


def test_bdf2_against_vode(backend):
    pc.check_bdf2_against_vode(backend)


def test_simulation_golden(backend):
    pc.check_simulation_golden(backend)


def test_resident_fields(backend):
    pc.check_resident_fields(backend)


def test_errors(backend):
    pc.check_errors(backend)


def test_heat_steady_state(backend):
    pc.check_heat_steady_state(backend, schemes.RODASPR, dirichlet=False)


def test_step_doubling_device_norm(backend):
    pc.check_step_doubling_device_norm(backend)


def test_fused_step_doubling(backend):
    pc.check_fused_step_doubling(backend)


def test_time_dependent_hook(backend):
    pc.check_time_dependent_hook(backend)


@pytest.mark.parametrize("name", sorted(pc.NOTEBOOK_CASES))
def test_notebook_models(name, backend):
    pc.check_notebook_model(name, backend)


def test_simulation_stays_resident(backend):
    pc.check_simulation_stays_resident(backend)


def test_container_on_device_fields(backend, tmp_path):
    pc.check_container_on_device_fields(backend, tmp_path)


def test_row_monitor(backend):
    pc.check_row_monitor(backend)


def test_constant_matrix_reuse(backend):
    pc.check_constant_matrix_reuse(backend)


def test_ensemble_restart(backend):
    pc.check_ensemble_restart(backend)


def test_bdf2_history_in_place(backend):
    pc.check_bdf2_history_in_place(backend)


def test_bdf2_history_is_the_hooked_state(backend):
    pc.check_bdf2_history_is_the_hooked_state(backend)


def test_hook_input_in_place(backend):
    pc.check_hook_input_in_place(backend)


def test_respike(backend):
    pc.check_respike(backend)


def test_fused_stage_rhs(backend):
    pc.check_fused_stage_rhs(backend)


def test_unstable_factorisation_recovers(backend):
    pc.check_unstable_factorisation_recovers(backend)


def test_ensemble_equals_single_members(backend):
    pc.check_ensemble_equals_single_members(backend, N=300, m1=8, m_upper=3)


def test_twenty_step_drift(backend):
    d = pc.drift_against_oracle(backend, 3, 400, "ROS2", nsteps=20, marks=(1, 20))
    assert d[1] <= 1e-11 and d[20] <= 1e-10, d


def test_python_hook_stays_resident(backend):
    pc.check_python_hook_stays_resident(backend)


def test_neumann_python_hook(backend):
    pc.check_neumann_python_hook(backend)


def test_adaptive_landing_reuse(backend):
    pc.check_adaptive_landing_reuse(backend)


def test_theta_bdf2_monitor(backend):
    pc.check_theta_bdf2_monitor(backend)
