"""Drop-in check with the reference's OWN classes (build container only: the
reference tree does not travel, so these tests skip on the GPU box).

`triflow_amd.compilers.hip_compiler` is handed to the reference's ``Model`` as
its ``compiler=`` callable (seam #1, model.py:152-155, 299-311), the reference's
own schemes then run on top of it, and the device schemes (seam #2) are driven
with the reference's ``Model`` and with a container class that is not ours.
The kernels run through the host emulation here; the GPU suite covers the HIP
build of the same code."""
from functools import partial

import numpy as np
import pytest

from oracle import corpus, ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def env():
    from tests.emu.build_emu import EmuBackend
    from triflow_amd.compilers import hip_compiler
    return ref_loader.load(), partial(hip_compiler, backend=EmuBackend())


def _pair(env, name):
    ref, plugin = env
    args = corpus.model_args(name)
    return ref.Model(*args, compiler="numpy"), ref.Model(*args, compiler=plugin)


@pytest.mark.parametrize("name", ["M1_advdiff", "M3_film", "M5_stiff", "helper_d", "upwind2_par", "bivar"])
@pytest.mark.parametrize("periodic", [True, False])
def test_plugin_inside_reference_model(env, name, periodic):
    m_ref, m_hip = _pair(env, name)
    assert type(m_hip).__module__ == "triflow.core.model"          # the reference's class
    fd = corpus.synthetic_fields(name, 48, seed=3, periodic=periodic)
    pars = corpus.synthetic_pars(name, 48, periodic)
    f_ref, f_hip = m_ref.fields_template(**fd), m_hip.fields_template(**fd)
    assert np.array_equal(m_ref.F(f_ref, pars), m_hip.F(f_hip, pars))
    J_ref, J_hip = m_ref.J(f_ref, pars), m_hip.J(f_hip, pars)
    assert J_hip.format == "csc" and J_hip.shape == J_ref.shape
    scale = abs(J_ref).max()
    assert abs(J_ref - J_hip).max() <= 4 * np.finfo(float).eps * scale   # duplicate-sum order only
    assert np.allclose(m_hip.J(f_hip, pars, sparse=False), J_ref.todense(), rtol=0, atol=1e-15 * scale)


@pytest.mark.parametrize("scheme", ["Theta", "ROS2", "ROS3PRL"])
def test_reference_schemes_on_the_plugin(env, scheme):
    """The reference's own time steppers, with F and J coming from the device path."""
    ref, _ = env
    m_ref, m_hip = _pair(env, "M1_advdiff")
    fd = corpus.synthetic_fields("M1_advdiff", 64, seed=5, periodic=False)
    pars = corpus.synthetic_pars("M1_advdiff", 64, False)
    kw = {} if scheme in ("Theta", "ROS2") else dict(time_stepping=False)
    s_ref, s_hip = getattr(ref.schemes, scheme)(m_ref, **kw), getattr(ref.schemes, scheme)(m_hip, **kw)
    f_ref, f_hip = m_ref.fields_template(**fd), m_hip.fields_template(**fd)
    t = 0.0
    for _ in range(3):
        _, f_ref = s_ref(t, f_ref, 0.3, pars, hook=corpus.dirichlet_hook_cfg1)
        t, f_hip = s_hip(t, f_hip, 0.3, pars, hook=corpus.dirichlet_hook_cfg1)
    assert np.abs(f_ref.uflat - f_hip.uflat).max() <= 1e-14


class ForeignFields:
    """A container that only offers the reference's protocol (fields.py:107-183):
    item access, ``copy``, ``fill``, ``uflat``, ``size``, ``dependent_variables``."""

    def __init__(self, dep, **arrays):
        self.dependent_variables = list(dep)
        self._data = {k: np.array(v, dtype=float) for k, v in arrays.items()}
        self.size = self._data["x"].size

    def __getitem__(self, key):
        return self._data[key]

    def copy(self):
        return ForeignFields(self.dependent_variables, **self._data)

    def fill(self, uflat):
        u = np.asarray(uflat).reshape(self.size, -1)
        for i, key in enumerate(self.dependent_variables):
            self._data[key][:] = u[:, i]

    @property
    def uflat(self):
        return np.vstack([self._data[k] for k in self.dependent_variables]).flatten("F")


@pytest.mark.parametrize("scheme", ["Theta", "ROS2", "RODASPR"])
def test_device_schemes_with_reference_model_and_foreign_fields(env, scheme):
    """Seam #2 from the other side: this package's GPU schemes given the reference's
    Model object and a container of a third party."""
    from triflow_amd import schemes as device_schemes
    ref, _ = env
    name = "M3_film"
    m_ref, m_hip = _pair(env, name)
    fd = corpus.synthetic_fields(name, 96, seed=2, periodic=True)
    pars = corpus.synthetic_pars(name, 96, True)
    kw = {} if scheme in ("Theta", "ROS2") else dict(time_stepping=False)
    s_ref = getattr(ref.schemes, scheme)(m_ref, **kw)
    s_dev = getattr(device_schemes, scheme)(m_hip, **kw)
    f_ref = m_ref.fields_template(**fd)
    f_dev = ForeignFields(m_hip._dep_vars, **fd)
    t = 0.0
    for _ in range(2):
        _, f_ref = s_ref(t, f_ref, 1e-3, pars)
        t, f_dev = s_dev(t, f_dev, 1e-3, pars)
    assert isinstance(f_dev, ForeignFields)
    assert np.abs(f_ref.uflat - f_dev.uflat).max() <= 1e-9 * np.abs(f_ref.uflat).max()
