"""Host-side front-end (Model, Fields): the behaviours the reference's own tests
pin (tests/test_model.py:118-194, tests/test_fields.py) that do not need a GPU."""
import pickle

import numpy as np
import pytest

from oracle import numpy_path as ora
from triflow_amd import Model
from triflow_amd.fields import BaseFields


def test_model_api_and_errors():
    model = Model(differential_equations=["k * dxxU + s"], dependent_variables="U",
                  parameters="k", help_functions="s", hold_compilation=True)
    assert set(model._args) == {"x", "U_m1", "U", "U_p1", "s_m1", "s", "s_p1", "k", "dx"}
    with pytest.raises(NotImplementedError):
        Model("dxxxxxU", "U", hold_compilation=True)
    with pytest.raises(ValueError):
        Model("dxxx(dx)", "U", hold_compilation=True)


@pytest.mark.parametrize("spelling", ["k * dxxU", "k * dx(dxU)", "k * dxx(U)",
                                      "k * Derivative(U, x, x)"])
def test_derivative_spellings(spelling):
    a = Model(spelling, "U", "k", hold_compilation=True)
    b = Model("k * dxxU", "U", "k", hold_compilation=True)
    assert [str(e) for e in a.F_array] == [str(e) for e in b.F_array]


def test_coercion_of_arguments():
    a = Model("k * dxxU", "U", "k", hold_compilation=True)
    b = Model(["k * dxxU"], ["U"], ["k"], hold_compilation=True)
    assert a._args == b._args and a._dep_vars == b._dep_vars == ("U",)


def test_compiler_seam_accepts_callables_and_strings():
    m = Model("k * dxxU", "U", "k", compiler=ora.numpy_compiler)
    x = np.linspace(0, 10, 50, endpoint=False)
    F = m.F(m.fields_template(x=x, U=np.cos(x)), dict(k=1., periodic=True))
    assert F.shape == (50,)
    for name in ("hip", "numpy", "theano"):      # one back end, three spellings
        Model("k * dxxU", "U", "k", compiler=name)
    with pytest.raises(ValueError):
        Model("k * dxxU", "U", "k", compiler="fortran")


def test_unsupported_expression_fails_at_compile_time():
    with pytest.raises(NotImplementedError):
        Model("gamma(U) * dxxU", "U")


def test_pickle_keeps_compiler_choice(tmp_path):
    m = Model("k * dxxT", "T", "k", compiler=ora.numpy_compiler)
    m.save(tmp_path / "heat")
    loaded = Model.load(tmp_path / "heat")
    x = np.linspace(0, 10, 50, endpoint=False)
    T = np.cos(x * 2 * np.pi / 10)
    pars = dict(periodic=True, k=1)
    assert loaded._symb_diff_eqs == m._symb_diff_eqs
    assert (loaded.J_array == m.J_array).all()
    assert loaded._args == m._args
    f1, f2 = m.fields_template(x=x, T=T), loaded.fields_template(x=x, T=T)
    assert (loaded.F(f2, pars) == m.F(f1, pars)).all()
    assert (loaded.J(f2, pars).todense() == m.J(f1, pars).todense()).all()


def test_routines_repr_and_diff_approx():
    m = Model("dxxU", "U", compiler=ora.numpy_compiler)
    x = np.linspace(0, 10, 30)
    fields = m.fields_template(x=x, U=np.cos(x * 2 * np.pi / 10))
    repr(m.F), repr(m.J), repr(m)
    J = m.J(fields, dict(periodic=True), sparse=False)
    Ja = m.F.diff_approx(fields, dict(periodic=True))
    assert np.allclose(Ja, J, rtol=1e-2, atol=1e-6)


# ---- Fields (reference tests/test_fields.py) ---------------------------------
def test_fields_uflat_fill_copy():
    F1 = BaseFields.factory1D(["U1", "U2"], ["s"])
    F2 = BaseFields.factory(("x",), [("U1", ("x",)), ("U2", ("x",))], [("s", ("x",))])
    x = np.linspace(0, 1, 20)
    U1, U2, s = np.cos(x), np.sin(x), x ** 2
    f = F1(x=x, U1=U1, U2=U2, s=s)
    g = F2(x=x, U1=U1, U2=U2, s=s)
    assert f.keys() == g.keys() == ["x", "U1", "U2", "s"]
    assert np.array_equal(f.uflat, np.vstack([U1, U2]).flatten("F"))
    assert f.size == 20 and f.dependent_variables == ["U1", "U2"]
    c = f.copy()
    c["U1"][0] = 42
    assert f["U1"][0] == U1[0]                     # deep copy
    f.fill(np.arange(40.0))
    assert np.array_equal(f["U1"], np.arange(0, 40, 2.0))
    assert np.array_equal(np.asarray(f.U2.values), np.arange(1, 40, 2.0))
    h = pickle.loads(pickle.dumps(f))
    assert h == f
    with pytest.raises(KeyError):
        F1(x=x, U1=U1)
    df = f.to_df()
    assert list(df.columns) == ["U1", "U2", "s"]


def test_fields_hook_idiom_writes_through():
    F = BaseFields.factory1D(["U"], [])
    f = F(x=np.linspace(0, 1, 5), U=np.zeros(5))
    f.U[0] = 1
    f["U"][-1] = 2
    assert f.uflat[0] == 1 and f.uflat[-1] == 2
    f["grad"] = np.gradient(f["U"])                # post-process style new entry
    assert "grad" in f


def test_block_size_limit_is_named_at_compile():
    """b = (stencil half width) x (variables) > 16 is refused when the model is compiled, with a
    message that names the limit (the reference accepts any size, model.py:138-150)."""
    import pytest
    from triflow_amd import Model
    from triflow_amd.codegen import UnsupportedExpression
    eqs = ["dxxxx%s + %s" % (v, w) for v, w in zip("ABCDGHKLM", "BCDGHKLMA")]
    with pytest.raises(UnsupportedExpression, match=r"b = 2 x 9 = 18"):
        Model(eqs, list("ABCDGHKLM"), None, None)
