"""TEST-ONLY alias package: lets the reference's own test files (which do
``from triflow import Model, Simulation, schemes``) run against triflow_amd.
The compiler behind every Model is the HIP plugin, executed through the host
emulation when no GPU is present (tests/emu) and through libtriflow_hip.so on a GPU."""
import os
from functools import partial

import triflow_amd
from triflow_amd import schemes, Simulation                     # noqa: F401
from triflow_amd.compilers import hip_compiler
from triflow_amd.container import TriflowContainer as Container, retrieve_container  # noqa: F401


def _backend():
    if os.environ.get("TRIFLOW_SHIM_BACKEND", "emu") == "hip":
        return None
    from tests.emu.build_emu import EmuBackend
    return EmuBackend()


_BACKEND = _backend()
device_compiler = hip_compiler if _BACKEND is None else partial(hip_compiler, backend=_BACKEND)


def _rebuild(eqs, dep, pars, helps, bdcs):
    return Model(eqs, dep, pars, helps, bdcs)


class Model(triflow_amd.Model):
    def __init__(self, *args, **kwargs):
        if kwargs.get("compiler", "theano") in ("theano", "numpy", "hip"):
            kwargs["compiler"] = device_compiler
        super().__init__(*args, **kwargs)

    def __reduce__(self):          # the emulation back end holds ctypes handles: rebuild by name
        return (_rebuild, (self._diff_eqs, self._dep_vars, self._pars, self._help_funcs, self._bdcs))
